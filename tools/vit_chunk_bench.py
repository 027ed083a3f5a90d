import sys, time, torch
sys.path.insert(0, '/root/repo')
from eavqa_amd.models.clip_vit import KNOWN_VITS, ClipVisionEncoder, random_init_vit_state_dict
dev = "cuda:0"
vcfg = KNOWN_VITS["ViT-L/14"]
vit = ClipVisionEncoder(vcfg, random_init_vit_state_dict(vcfg, 2021, dev), torch.bfloat16, dev)
px = torch.randn(160, 3, vcfg.image, vcfg.image, device=dev)
def run(chunk):
    outs = [vit.encode_image(px[i:i + chunk]) for i in range(0, 160, chunk)]
    return torch.cat(outs, 0)
ref = run(160)
for chunk in (160, 80, 64, 40, 32, 16):
    for _ in range(2): run(chunk)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): out = run(chunk)
    e1.record(); torch.cuda.synchronize()
    print(f"chunk {chunk:4d}: {e0.elapsed_time(e1) / 5:7.2f} ms   max|diff vs 160| {(out.float() - ref.float()).abs().max().item():.2e}", flush=True)
