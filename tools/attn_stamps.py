#!/usr/bin/env python3
"""Phase times inside attn_decode_kernel (workgroup (0, 0), per wave) from the profiling build of tools/attn_stamps.sh.
Cases: the OPT-2.7B few-shot decode step (B 32, 32 heads x 80, 160 keys, q | k | v from 4 partial-sum slices), a T0-3B self-attention step
(5 keys, relative bias) and cross-attention step (150 keys, q from partial sums).  Caches rotated: cold HBM."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "_bin", "libeavqa_attn_stamps.so"))
i32, i64, f32, ptr = C.c_int, C.c_int64, C.c_float, C.c_void_p
lib.eavqa_attention_decode_splitk_rel.argtypes = [i32, i32, i32, i32, i32, ptr, i32, i32, ptr, i64, ptr, i64, i64, ptr, i64, ptr, i64, f32, ptr, i64, i32, ptr]
lib.eavqa_attn_stamps_read.argtypes = [ptr]
dev, bf = "cuda", torch.bfloat16
NAMES = ["V DMA + K batch issued", "new K/V row summed + appended", "(q issue)", "q summed", "K scored", "(V landed)", "barrier", "max / sum",
         "P.V", "(reduce)", "partials + store"]

def case(tag, B, H, hd, Sk, ks, cols_mult, rel, masked):
    I = H * hd
    S_max = Sk + 9
    n = 12
    caches = [(torch.randn(B * S_max, I, device=dev).to(bf), torch.randn(B * S_max, I, device=dev).to(bf)) for _ in range(n)]
    part = torch.randn(ks, B, cols_mult * I, device=dev) * 0.3
    o = torch.empty(B, I, device=dev, dtype=bf)
    relt = torch.randn(H, 2 * Sk + 1, device=dev) if rel else None
    km = torch.ones(B, Sk, dtype=torch.int32, device=dev) if masked else None
    stream = torch.cuda.current_stream().cuda_stream
    def run(i):
        k, v = caches[i % n]
        rc = lib.eavqa_attention_decode_splitk_rel(1, B, H, Sk, hd, part.data_ptr(), ks, cols_mult * I, k.data_ptr(), I, v.data_ptr(), I, S_max, o.data_ptr(), I,
                                                   km.data_ptr() if km is not None else None, Sk if km is not None else 0, 1.0,
                                                   relt.data_ptr() if relt is not None else None, relt.stride(0) if relt is not None else 0, Sk, stream)
        assert rc == 0, rc
    for i in range(3): run(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(24): run(i)
    e1.record(); torch.cuda.synchronize()
    buf = (C.c_ulonglong * 256)()
    assert lib.eavqa_attn_stamps_read(buf) == 0
    print(f"== {tag}: B {B}, {H} heads x {hd}, {Sk} keys, ks {ks}: {e0.elapsed_time(e1) / 24 * 1e3:.1f} us per launch (events, back to back)")
    waves = 16
    t0 = min(buf[w * 16] for w in range(waves) if buf[w * 16])
    for w in (0, 1, 5, 15):
        st = [buf[w * 16 + i] for i in range(10)]
        if not st[0]:
            continue
        d = [(st[i] - t0) * 0.01 for i in range(10)]
        print(f"  wave {w:2d}: " + "  ".join(f"[{i}] {d[i]:5.2f}" for i in range(10)) + "  us since the workgroup's first stamp")
    last = max(buf[w * 16 + 9] for w in range(waves))
    print(f"  workgroup (0,0) alive for {(last - t0) * 0.01:.2f} us; stamps: 0 start, 1 loads issued, 2 new row appended, 3 q ready, 4 scored, 5 V landed, 6 barrier, 7 softmax stats, 8 P.V done, 9 end")

case("OPT-2.7B decode step", 32, 32, 80, 160, 4, 3, False, True)
case("T0-3B self-attention step", 32, 32, 64, 5, 4, 3, True, False)
case("T0-3B cross-attention step", 32, 32, 64, 150, 8, 1, False, True)
