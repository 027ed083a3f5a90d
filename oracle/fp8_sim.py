"""fp8 numerics model for the oracle (TEST INFRASTRUCTURE, like everything under oracle/).

The HIP fp8 path (BASELINE configs[4]; csrc/gemm_fp8.hip, models/lm.py ``Fp8Weight``) multiplies e4m3 weights (one float scale per
tensor) by activations that are bf16 in memory and quantised to e4m3 ROW by ROW (scale = amax / 448) in front of every GEMM, forward
and backward, accumulating in fp32.  ``fp8_linear`` below restates exactly that dataflow in torch on the CPU as a differentiable
function; plugged into ``oracle.ref_cpu`` through ``cfg["linear_fn"]`` it turns the fp32 oracle into a model of the fp8 LM, so that
a comparison with the HIP path sees only what a kernel may legitimately differ in (accumulation order; an e4m3 rounding that flips
because an upstream bf16 value differed in its last bit) and not the quantisation noise itself.

The judge's comparison (fp32 oracle with the WEIGHTS round-tripped through e4m3, activations exact) is kept beside it in the tests:
that one bounds what the fp8 format costs end to end.
"""
import torch


def quantize_rows(x: torch.Tensor):
    """``x[..., :]`` -> (values representable in e4m3 after dividing by the row scale, row scale) - eavqa_quantize_rows_fp8."""
    amax = x.abs().amax(-1, keepdim=True)
    scale = torch.where(amax > 0, amax * torch.tensor(1.0 / 448.0, dtype=torch.float32), torch.ones_like(amax))
    q = (x * (1.0 / scale)).to(torch.float8_e4m3fn).float()
    return q, scale


def fake_quant_rows(x: torch.Tensor) -> torch.Tensor:
    """Row-wise e4m3 round trip of the bf16 image of ``x`` (the GEMM operand lives in HBM as bf16)."""
    q, scale = quantize_rows(x.bfloat16().float())
    return q * scale


class _Fp8Linear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w_rt):
        ctx.save_for_backward(w_rt)
        return fake_quant_rows(x) @ w_rt.T

    @staticmethod
    def backward(ctx, dy):
        (w_rt,) = ctx.saved_tensors
        shape = dy.shape
        d2 = dy.reshape(-1, shape[-1])
        return (fake_quant_rows(d2) @ w_rt).reshape(*shape[:-1], w_rt.shape[1]), None      # dgrad only: the LM is frozen


def fp8_linear(x: torch.Tensor, w_roundtripped: torch.Tensor) -> torch.Tensor:
    """``cfg["linear_fn"]``: ``w_roundtripped`` is the [out, in] weight after its per-tensor e4m3 round trip."""
    return _Fp8Linear.apply(x, w_roundtripped)
