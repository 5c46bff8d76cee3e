"""PyTorch restatement of the reference's grid_sample fallback.  TEST INFRASTRUCTURE ONLY.

Follows models/csrc/wrapper.py:14-38: per level a 5-D ``F.grid_sample`` (trilinear over
(view, y, x), zeros padding, align_corners=True) on channel-first ``[B, C, N, H, W]``
features, multiplied by that level's scale weight and summed.  It is the CPU baseline the
benchmark times on the host cores, and a second checker beside the C oracle.

Differences from the device kernels that tests must account for
(msmv_sampling_forward.cu:110, msmv_sampling_backward.cu:103-104):
  * the view axis is interpolated here but rounded to the nearest view on the device;
    identical for locations of the form view/(N-1);
  * the gradient of the view coordinate is non-zero here and exactly 0 on the device.
"""
import torch.nn.functional as F


def to_channel_first(feats_cl):
    """[B, N, H, W, C] -> [B, C, N, H, W] (the layout the fallback takes)."""
    return [f.permute(0, 4, 1, 2, 3).contiguous() for f in feats_cl]


def grid_sample_ref(feats_cf, loc, weights):
    """feats_cf: list of [B,C,N,H,W]; loc [B,Q,P,3] in [0,1]; weights [B,Q,P,L] -> [B,Q,C,P]."""
    assert weights.shape[-1] == len(feats_cf)
    B, C = feats_cf[0].shape[:2]
    _, Q, P, _ = loc.shape
    grid = (loc * 2 - 1)[:, :, :, None, :]          # wrapper.py:25-26
    acc = None
    for lvl, feat in enumerate(feats_cf):
        s = F.grid_sample(feat, grid, mode='bilinear', padding_mode='zeros',
                          align_corners=True)[..., 0]          # wrapper.py:31-34
        s = s * weights[..., lvl].reshape(B, 1, Q, P)           # wrapper.py:35
        acc = s if acc is None else acc + s
    return acc.permute(0, 2, 1, 3)                              # wrapper.py:38


def grid_sample_ref_channel_last(feats_cl, loc, weights):
    """Same, taking the device layout [B,N,H,W,C]."""
    return grid_sample_ref(to_channel_first(feats_cl), loc, weights)
