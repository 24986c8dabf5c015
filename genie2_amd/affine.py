"""Rigid frames `T` (rots [...,3,3], trans [...,3]) -- the value type the
reference passes across its seam (genie/utils/affine_utils.py:54-244).  Host-side
convenience only; the frame algebra of the hot path runs inside the HIP kernels."""
import torch


def _rot_apply(r, v):
    return torch.stack([(r[..., i, :] * v).sum(-1) for i in range(3)], dim=-1)


class T:
    def __init__(self, rots, trans):
        if rots is None and trans is None:
            raise ValueError('Only one of rots and trans can be None')
        if rots is None:
            rots = torch.eye(3, dtype=trans.dtype, device=trans.device).expand(*trans.shape[:-1], 3, 3)
        if trans is None:
            trans = torch.zeros(*rots.shape[:-2], 3, dtype=rots.dtype, device=rots.device)
        if rots.shape[-2:] != (3, 3) or trans.shape[-1] != 3 or rots.shape[:-2] != trans.shape[:-1]:
            raise ValueError('Incorrectly shaped input')
        self.rots, self.trans = rots, trans

    @property
    def shape(self):
        s = self.rots.shape[:-2]
        return s if len(s) > 0 else torch.Size([1])

    def __getitem__(self, index):
        index = index if isinstance(index, tuple) else (index,)
        return T(self.rots[index + (slice(None), slice(None))], self.trans[index + (slice(None),)])

    def get_trans(self):
        return self.trans

    def get_rots(self):
        return self.rots

    def scale_translation(self, factor):
        return T(self.rots, self.trans * factor)

    def compose(self, other):
        return T(torch.matmul(self.rots, other.rots), _rot_apply(self.rots, other.trans) + self.trans)

    def apply(self, pts):
        return _rot_apply(self.rots, pts) + self.trans

    def invert_apply(self, pts):
        return _rot_apply(self.rots.transpose(-1, -2), pts - self.trans)

    def to(self, device):
        return T(self.rots.to(device), self.trans.to(device))
