"""The tfluids nn.Modules as torch.nn.Modules with autograd, on the HIP operators.

Host mirror of torch/tfluids/{velocity_divergence,velocity_update,set_wall_bcs,flags_to_occupancy,
volumetric_up_sampling_nearest}.lua: the reference wraps each native op in an nn.Module whose updateOutput /
updateGradInput call the Forward / Backward entry points; here each is a torch.autograd.Function over the same entry
points of libtfluids_hip.so plus a thin nn.Module of the reference's name and call shape (inputs in the reference's
table order). Like the reference, no gradient flows to `flags` (its gradInput is zero-filled there, None here).
Everything is fp32 on an MI355X; there is no CPU fallback.
"""
import torch

from . import tfluids


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


class _VelocityDivergenceFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, U, flags):                       # velocity_divergence.lua:28-37
        U, flags = _c(U), _c(flags)
        out = torch.empty_like(flags)
        tfluids.velocityDivergenceForward(U, flags, out)
        ctx.save_for_backward(U, flags)
        return out

    @staticmethod
    def backward(ctx, grad_out):                      # velocity_divergence.lua:39-50
        U, flags = ctx.saved_tensors
        gradU = torch.empty_like(U)
        tfluids.velocityDivergenceBackward(U, flags, _c(grad_out), gradU)
        return gradU, None


class _VelocityUpdateFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, p, U, flags):                    # velocity_update.lua:29-39: output = copy of U, updated in place
        p, U, flags = _c(p), _c(U), _c(flags)
        out = U.clone()
        tfluids.velocityUpdateForward(out, flags, p)
        ctx.save_for_backward(p, U, flags)
        return out

    @staticmethod
    def backward(ctx, grad_out):                      # velocity_update.lua:41-55: gradU is ZERO in the reference
        p, U, flags = ctx.saved_tensors
        gradP = torch.empty_like(p)
        tfluids.velocityUpdateBackward(U, flags, p, _c(grad_out), gradP)
        return gradP, torch.zeros_like(U), None


class _SetWallBcsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, U, flags):                       # set_wall_bcs.lua:29-48 (output = U * mask)
        U, flags = _c(U), _c(flags)
        out = U.clone()
        tfluids.setWallBcsForward(out, flags)
        ctx.save_for_backward(flags)
        return out

    @staticmethod
    def backward(ctx, grad_out):                      # set_wall_bcs.lua:50-66 (gradU = mask * gradOutput)
        (flags,) = ctx.saved_tensors
        return tfluids.setWallBcsBackward(flags, _c(grad_out)), None


class _UpSamplingNearestFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, ratio):                       # volumetric_up_sampling_nearest.lua:27-37
        x = _c(x)
        B, C, Z, Y, X = x.shape
        out = x.new_empty(B, C, Z * ratio, Y * ratio, X * ratio)
        tfluids.volumetricUpSamplingNearestForward(ratio, x, out)
        ctx.save_for_backward(x)
        ctx.ratio = ratio
        return out

    @staticmethod
    def backward(ctx, grad_out):                      # volumetric_up_sampling_nearest.lua:39-48
        (x,) = ctx.saved_tensors
        gradIn = torch.empty_like(x)
        tfluids.volumetricUpSamplingNearestBackward(ctx.ratio, x, _c(grad_out), gradIn)
        return gradIn, None


class VelocityDivergence(torch.nn.Module):
    """tfluids.VelocityDivergence: forward({U, flags}) -> div."""

    def forward(self, inputs):
        U, flags = inputs
        return _VelocityDivergenceFn.apply(U, flags)


class VelocityUpdate(torch.nn.Module):
    """tfluids.VelocityUpdate: forward({p, U, flags}) -> U - grad(p) (Manta's correctVelocity)."""

    def forward(self, inputs):
        p, U, flags = inputs
        return _VelocityUpdateFn.apply(p, U, flags)


class SetWallBcs(torch.nn.Module):
    """tfluids.SetWallBcs: forward({U, flags}) -> U with the wall-normal components of obstacle faces zeroed."""

    def forward(self, inputs):
        U, flags = inputs
        return _SetWallBcsFn.apply(U, flags)


class FlagsToOccupancy(torch.nn.Module):
    """tfluids.FlagsToOccupancy: Manta flags -> {0, 1} occupancy; no gradient (flags_to_occupancy.lua:33-36)."""

    def forward(self, flags):
        flags = _c(flags)
        out = torch.empty_like(flags)
        tfluids.flagsToOccupancy(flags.detach(), out)
        return out


class VolumetricUpSamplingNearest(torch.nn.Module):
    """tfluids.VolumetricUpSamplingNearest(ratio) on [B, C, Z, Y, X]."""

    def __init__(self, ratio):
        super().__init__()
        if int(ratio) != ratio or ratio <= 0:
            raise tfluids.TfluidsError("ratio must be a non-zero positive integer")
        self.ratio = int(ratio)

    def forward(self, x):
        tfluids._check(x.dim() == 5, "Only batch mode is supported for now.")
        return _UpSamplingNearestFn.apply(x, self.ratio)

    def extra_repr(self):
        return "ratio=%d" % self.ratio
