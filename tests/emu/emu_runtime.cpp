// TEST INFRASTRUCTURE: fiber scheduler behind tests/emu/hip/hip_runtime.h.
#include <hip/hip_runtime.h>

namespace emu {
Block *B = nullptr;
EmuDim3 g_grid = {1, 1, 1};
static const size_t STACK = 256 * 1024;

static void trampoline() {
    Block *b = B;
    b->entry(b->entry_arg);
    Lane &l = b->lanes[b->cur];
    l.done = true;
    swapcontext(&l.ctx, &b->sched);
}

uint64_t park(int op, uint64_t arg, int arg2) {
    Lane &l = me();
    l.op = op;
    l.arg = arg;
    l.arg2 = arg2;
    l.parked = true;
    swapcontext(&l.ctx, &B->sched);
    return l.res;
}

static void die(const char *msg, int wave) {
    fprintf(stderr, "[emu] FATAL: %s (wave %d)\n", msg, wave);
    abort();
}

void run_block(void (*entry)(void *), void *arg, int block_idx, int nthreads, int block_idy, int block_idz) {
    static Block blk;
    static Lane *lanes = nullptr;
    static int cap = 0;
    if (cap < nthreads) {
        lanes = (Lane *)calloc(nthreads, sizeof(Lane));
        for (int i = 0; i < nthreads; i++) lanes[i].stack = (char *)malloc(STACK);
        cap = nthreads;
    }
    B = &blk;
    blk.nthreads = nthreads;
    blk.lanes = lanes;
    blk.bidx = EmuDim3{block_idx, block_idy, block_idz};
    blk.bdim = EmuDim3{nthreads, 1, 1};
    blk.entry = entry;
    blk.entry_arg = arg;
    for (int i = 0; i < nthreads; i++) {
        Lane &l = lanes[i];
        l.tid = i;
        l.done = false;
        l.parked = false;
        l.op = OP_NONE;
        getcontext(&l.ctx);
        l.ctx.uc_stack.ss_sp = l.stack;
        l.ctx.uc_stack.ss_size = STACK;
        l.ctx.uc_link = nullptr;
        makecontext(&l.ctx, (void (*)())trampoline, 0);
    }
    const int nwaves = (nthreads + 63) / 64;
    for (;;) {
        bool progress = false, all_done = true;
        // run every runnable lane until it parks or finishes
        for (int i = 0; i < nthreads; i++) {
            Lane &l = lanes[i];
            if (l.done || l.parked) continue;
            blk.cur = i;
            swapcontext(&blk.sched, &l.ctx);
            progress = true;
        }
        // resolve wave-level operations
        for (int w = 0; w < nwaves; w++) {
            int lo = w * 64, hi = lo + 64 > nthreads ? nthreads : lo + 64;
            int op = -1, live = 0;
            bool same = true;
            for (int i = lo; i < hi; i++) {
                if (lanes[i].done) continue;
                live++;
                if (!lanes[i].parked) { same = false; break; }
                if (op < 0) op = lanes[i].op;
                else if (op != lanes[i].op) die("lanes of a wave parked at different cross-lane ops (divergent use)", w);
            }
            if (!live || !same || op == OP_SYNCTHREADS) continue;
            // a wave-level op needs every live lane of the wave; lanes that exited earlier would hang a real wave too
            for (int i = lo; i < hi; i++)
                if (lanes[i].done) die("cross-lane op after some lanes of the wave exited", w);
            if (op == OP_BALLOT) {
                uint64_t m = 0;
                for (int i = lo; i < hi; i++) if (lanes[i].arg) m |= 1ull << (i - lo);
                for (int i = lo; i < hi; i++) lanes[i].res = m;
            } else if (op == OP_MFMA32) {
                if (hi - lo != 64) die("MFMA needs a full wave", w);
                float A[32][2], Bm[2][32];
                MfmaOperands *o[64];
                for (int l = 0; l < 64; l++) {
                    o[l] = (MfmaOperands *)(uintptr_t)lanes[lo + l].arg;
                    A[l & 31][l >> 5] = o[l]->a;
                    Bm[l >> 5][l & 31] = o[l]->b;
                }
                for (int l = 0; l < 64; l++)
                    for (int r = 0; r < 16; r++) {
                        const int i = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), j = l & 31;
                        o[l]->acc[r] = fmaf(A[i][1], Bm[1][j], fmaf(A[i][0], Bm[0][j], o[l]->acc[r]));
                    }
            } else if (op == OP_SHFL) {
                uint64_t vals[64];
                for (int i = lo; i < hi; i++) vals[i - lo] = lanes[i].arg;
                for (int i = lo; i < hi; i++) lanes[i].res = vals[lanes[i].arg2];
            }
            for (int i = lo; i < hi; i++) lanes[i].parked = false;
            progress = true;
        }
        // block barrier
        {
            int live = 0, at = 0;
            for (int i = 0; i < nthreads; i++) {
                if (lanes[i].done) continue;
                live++;
                if (lanes[i].parked && lanes[i].op == OP_SYNCTHREADS) at++;
            }
            if (live && at == live) {
                for (int i = 0; i < nthreads; i++) lanes[i].parked = false;
                progress = true;
            }
        }
        for (int i = 0; i < nthreads; i++) all_done &= lanes[i].done;
        if (all_done) break;
        if (!progress) die("deadlock: no lane can make progress", -1);
    }
}
}  // namespace emu
