"""Actor-critic networks of the hot path's callers, kept in PyTorch-ROCm (north_star: the ResNet/SE tower
stays in PyTorch so EnvManager/PPOAgent/Trainer stay drop-in).  Parameter names match the reference's
state dicts (keisei/core/neural_network.py:10-29, keisei/training/models/resnet_tower.py:15-83) so its
checkpoints load: stem, bn_stem, res_blocks.{i}.{conv1,bn1,conv2,bn2,se.fc1,se.fc2}, policy_head.{0,1,4},
value_head.{0,1,4}.  forward(obs) -> (policy_logits [B,13527], value [B]); the masked soft-max / sampling /
loss tail is NOT here -- it runs in the HIP kernels (shogidrl_amd.ops)."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import tower
from ._lib import NUM_ACTIONS


class TinyActorCritic(nn.Module):
    """conv3x3(46->16) + two Linear heads on 16*81 features (reference ActorCritic)."""

    def __init__(self, input_channels=46, num_actions_total=NUM_ACTIONS):
        super().__init__()
        self.conv = nn.Conv2d(input_channels, 16, kernel_size=3, padding=1)
        self.policy_head = nn.Linear(16 * 81, num_actions_total)
        self.value_head = nn.Linear(16 * 81, 1)

    @property
    def policy_linear(self):
        """the last layer of the policy head: the fused head kernels take its weight / bias and its INPUT features"""
        return self.policy_head

    def policy_features(self, x):
        """(input of policy_linear [B,1296] float32, value [B]): forward() without the last Linear of the policy head"""
        h = torch.flatten(F.relu(self.conv(x)), 1).float()
        return h, self.value_head(h).squeeze(-1)

    def forward(self, x):
        h = torch.flatten(F.relu(self.conv(x)), 1)
        return self.policy_head(h), self.value_head(h).squeeze(-1)


FUSED_TOWER_GLUE = True      # False: always take the ATen path (tests compare the two)
MERGED_BN_SE = True          # False: second batch norm and SE tail as two separate fused nodes


class _BN(nn.BatchNorm2d):
    """nn.BatchNorm2d (same parameters / buffers / state-dict keys) with the following ReLU / residual add folded in:
    `forward(x, residual=None, relu=False)` = [relu](bn(x) [+ residual]).  Channels-last CUDA activations go through
    the fused HIP kernels (shogidrl_amd.tower: one statistics pass + one apply pass instead of the framework's
    statistics / transform / add / relu kernels); everything else (CPU, the 2-plane heads) through ATen's native batch
    norm -- not MIOpen's, which on MI355X pairs worse with the NHWC implicit-GEMM convolutions."""

    def forward(self, x, residual=None, relu=False, pre_bias=None):
        if FUSED_TOWER_GLUE and x.is_cuda:
            y = tower.bn_act(x, self, residual, relu, pre_bias)
            if y is not None:
                return y
        if pre_bias is not None:
            x = x + pre_bias.view(1, -1, 1, 1).to(x.dtype)
        if x.is_cuda:
            with torch.backends.cudnn.flags(enabled=False):
                y = super().forward(x)
        else:
            y = super().forward(x)
        if residual is not None:
            y = y + residual
        return F.relu(y) if relu else y


def _conv_bn(conv, bn, x, residual=None, relu=False):
    """[relu](bn(conv(x)) [+ residual]).  On the GPU the convolution runs without its bias, which is handed to the
    fused batch norm instead (it only moves the running mean): that removes the bias-add pass after every convolution
    and the bias-gradient reduction over the whole activation in the backward."""
    if FUSED_TOWER_GLUE and x.is_cuda and conv.bias is not None:
        h = F.conv2d(x, conv.weight, None, conv.stride, conv.padding, conv.dilation, conv.groups)
        return bn(h, residual, relu, conv.bias)
    return bn(conv(x), residual, relu)


class _SE(nn.Module):
    def __init__(self, channels, ratio):
        super().__init__()
        hidden = max(1, int(channels * ratio))
        self.fc1 = nn.Conv2d(channels, hidden, 1)
        self.fc2 = nn.Conv2d(hidden, channels, 1)

    def forward(self, x):
        s = x.mean(dim=(2, 3), keepdim=True)
        return x * torch.sigmoid(self.fc2(F.relu(self.fc1(s))))


class _Block(nn.Module):
    def __init__(self, channels, se_ratio):
        super().__init__()
        self.conv1 = nn.Conv2d(channels, channels, 3, padding=1)
        self.bn1 = _BN(channels)
        self.conv2 = nn.Conv2d(channels, channels, 3, padding=1)
        self.bn2 = _BN(channels)
        self.se = _SE(channels, se_ratio) if se_ratio else None

    def forward(self, x):
        y = _conv_bn(self.conv1, self.bn1, x, relu=True)
        if self.se is None:
            return _conv_bn(self.conv2, self.bn2, y, residual=x, relu=True)
        if FUSED_TOWER_GLUE and MERGED_BN_SE and y.is_cuda and self.conv2.bias is not None and self.bn2.training:
            # second batch norm + squeeze-excitation tail as one autograd node (two HBM passes fewer)
            c = self.conv2
            h = F.conv2d(y, c.weight, None, c.stride, c.padding, c.dilation, c.groups)
            z = tower.bn_se_tail(h, self.bn2, self.se, x, c.bias)
            if z is not None:
                return z
            y = self.bn2(h, pre_bias=c.bias)
        else:
            y = _conv_bn(self.conv2, self.bn2, y)
        z = tower.se_tail(y, self.se, x) if (FUSED_TOWER_GLUE and y.is_cuda) else None
        return z if z is not None else F.relu(self.se(y) + x)


class ResTowerActorCritic(nn.Module):
    """ResNet tower + optional squeeze-excitation, slim 2-plane policy / value heads (reference
    ActorCriticResTower; default 9 blocks x 256 channels, SE 0.25 = 13 241 300 parameters)."""

    def __init__(self, input_channels=46, num_actions_total=NUM_ACTIONS, tower_depth=9, tower_width=256, se_ratio=0.25):
        super().__init__()
        self.stem = nn.Conv2d(input_channels, tower_width, 3, padding=1)
        self.bn_stem = _BN(tower_width)
        self.res_blocks = nn.Sequential(*[_Block(tower_width, se_ratio) for _ in range(tower_depth)])
        self.policy_head = nn.Sequential(nn.Conv2d(tower_width, 2, 1), _BN(2), nn.ReLU(), nn.Flatten(),
                                         nn.Linear(2 * 81, num_actions_total))
        self.value_head = nn.Sequential(nn.Conv2d(tower_width, 2, 1), _BN(2), nn.ReLU(), nn.Flatten(),
                                        nn.Linear(2 * 81, 1))

    def trunk(self, x):
        return self.res_blocks(_conv_bn(self.stem, self.bn_stem, x, relu=True))

    @property
    def policy_linear(self):
        """nn.Linear(2*9*9, 13527) (resnet_tower.py:63-69): the layer the fused head kernels absorb"""
        return self.policy_head[4]

    def policy_features(self, x):
        """(input of policy_linear [B,162] float32, value [B]) -- forward() minus the policy head's last Linear; the
        agent hands these features to ops.policy_head_* so that the [B,13527] logits are never materialised"""
        h = self.trunk(x)
        p, v = self.policy_head[0](h), self.value_head[0](h)
        with torch.autocast(device_type=x.device.type, enabled=False):
            p, v = p.float(), v.float()
            for layer in list(self.policy_head)[1:4]:
                p = layer(p)
            for layer in list(self.value_head)[1:]:
                v = layer(v)
            return p, v.squeeze(-1)

    def forward(self, x):
        h = self.trunk(x)
        # the 1x1 convolutions that squeeze the trunk to 2 planes follow the ambient autocast (they read the 170 MB
        # trunk output; in float32 that costs a 340 MB cast and float32 convolutions for 2 output channels); from the
        # 2-plane features on -- batch norm, ReLU and the Linear layers that produce the logits and the value, which
        # feed the soft-max / PPO-loss kernels -- everything is float32 even under bf16 autocast
        p, v = self.policy_head[0](h), self.value_head[0](h)
        with torch.autocast(device_type=x.device.type, enabled=False):
            p, v = p.float(), v.float()
            for layer in list(self.policy_head)[1:]:
                p = layer(p)
            for layer in list(self.value_head)[1:]:
                v = layer(v)
            return p, v.squeeze(-1)


def model_factory(model_type, obs_shape=(46, 9, 9), num_actions=NUM_ACTIONS, tower_depth=9, tower_width=256,
                  se_ratio=0.25, **_unused):
    """Same dispatch keys as keisei/training/models/__init__.py:6 (plus the tiny CNN)."""
    if model_type in ("resnet", "dummy", "testmodel", "resumemodel"):
        if model_type != "resnet":
            tower_depth, tower_width, se_ratio = 1, 16, None
        return ResTowerActorCritic(obs_shape[0], num_actions, tower_depth, tower_width, se_ratio)
    if model_type in ("tiny", "cnn"):
        return TinyActorCritic(obs_shape[0], num_actions)
    raise ValueError("Unknown model_type: %s" % model_type)
