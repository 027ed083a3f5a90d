#!/usr/bin/env python3
"""Few-shot generate throughput with the NEXT batch's ViT encode issued on a second stream while the current batch is generated
(prefill + decode), against the sequential loop of bench.py's few-shot leg.  OPT-2.7B (cfg4) shapes.

    python tools/fewshot_pipeline_probe.py [--reps 6] [--priority]
"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=6)
    a = ap.parse_args()
    from eavqa_amd.data.synthetic import fewshot_batch
    from eavqa_amd.models.clip_vit import KNOWN_VITS, ClipVisionEncoder, random_init_vit_state_dict
    from eavqa_amd.models.clipcap import ClipCaptionPrefix
    from eavqa_amd.models.lm import KNOWN_CONFIGS, FrozenCausalLM, LMConfig, random_init_state_dict
    device, dtype = "cuda:0", torch.bfloat16
    f = bench.FEWSHOT
    vcfg = KNOWN_VITS[f["vit"]]
    lcfg = LMConfig.from_hf_dict(KNOWN_CONFIGS[f["lm"]])
    vit = ClipVisionEncoder(vcfg, random_init_vit_state_dict(vcfg, 2021, device), dtype, device)
    lm = FrozenCausalLM(lcfg, random_init_state_dict(lcfg, 2021, device), dtype, device)
    torch.manual_seed(2021)
    model = ClipCaptionPrefix(prefix_length=f["prefix_length"], prefix_size=vcfg.proj, mapping_type="mlp", lm=lm, dtype=dtype, device=device).eval()
    sentinel = lcfg.vocab - 1
    b = fewshot_batch(f["batch"], lcfg.vocab, f["shots"], f["seg_len"], sentinel, image_size=vcfg.image, device=device)
    B, n_img = f["batch"], f["shots"] + 1
    px = b["pixel_values"].reshape(B * n_img, *b["pixel_values"].shape[2:])

    def gen(emb):
        return model.generate_fewshot(b["input_ids"], emb, b["attention_mask"], num_shots=f["shots"], special_token_id=sentinel,
                                      max_length=f["new_tokens"], pad_token_id=lcfg.pad_token_id, eos_token_id=None)

    def sequential(reps):
        for _ in range(reps):
            out = gen(vit.encode_image(px).view(B, n_img, -1))
        return out

    def pipelined(reps, side):
        main_s = torch.cuda.current_stream()
        def submit():
            side.wait_stream(main_s)                    # (inputs of the encode were produced on the main stream)
            with torch.cuda.stream(side):
                e = vit.encode_image(px).view(B, n_img, -1)
                ev = torch.cuda.Event()
                ev.record(side)
            return e, ev
        nxt = submit()
        for i in range(reps):
            emb, ev = nxt
            main_s.wait_event(ev)
            emb.record_stream(main_s)
            if i + 1 < reps:
                nxt = submit()
            out = gen(emb)
        return out

    try:
        lo, hi = torch.cuda.Stream.priority_range()
    except Exception:
        lo, hi = 0, -1
    LOW = max(lo, hi)
    print("stream priority range (least, greatest):", lo, hi, flush=True)

    def hi_main(reps):
        s = torch.cuda.Stream(priority=min(lo, hi))
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            out = pipelined(reps, torch.cuda.Stream(priority=LOW))
        torch.cuda.current_stream().wait_stream(s)
        return out

    ref = sequential(1)
    torch.cuda.synchronize()
    for name, fn in (("sequential", lambda: sequential(a.reps)),
                     ("pipelined, equal priority", lambda: pipelined(a.reps, torch.cuda.Stream())),
                     ("pipelined, encode on the lowest-priority stream", lambda: pipelined(a.reps, torch.cuda.Stream(priority=LOW))),
                     ("pipelined, generation on the highest-priority stream", lambda: hi_main(a.reps)),
                     ("sequential", lambda: sequential(a.reps))):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / a.reps
        same = out == ref
        print(f"{name:48s} {dt * 1e3:8.2f} ms per batch  {B / dt:8.1f} questions/s   ids equal to the sequential run: {same}", flush=True)


if __name__ == "__main__":
    main()
