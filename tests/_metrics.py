"""Gradient comparison statistics shared by the parity tests: direction (cosine), length (norm ratio) and the largest entry.

A bound on the max-relative error of the largest entry alone cannot tell a wrong gradient from noise (a zero gradient scores 1.0);
cosine similarity and the norm ratio of the WHOLE gradient can: zero -> cosine undefined / ratio 0, sign-flipped -> cosine -1,
mis-scaled -> ratio off."""
import torch


def grad_stats(got: dict, ref: dict):
    """(cosine, |got| / |ref|, max over tensors of max|got - ref| / max|ref|) over the parameters named in ``ref``."""
    g = torch.cat([got[k].detach().double().flatten().cpu() for k in ref])
    r = torch.cat([ref[k].detach().double().flatten().cpu() for k in ref])
    gn, rn = g.norm().item(), r.norm().item()
    cos = (torch.dot(g, r).item() / (gn * rn)) if gn > 0 and rn > 0 else 0.0
    maxrel = max((got[k].detach().double().cpu() - ref[k].detach().double().cpu()).abs().max().item() / max(ref[k].detach().abs().max().item(), 1e-30)
                 for k in ref)
    return cos, (gn / rn if rn > 0 else float("inf")), maxrel
