import cProfile, pstats, pathlib, shutil, sys, os, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sd_mecha_amd as amd
import bench_e2e as B
from sd_mecha_amd.shapes import load_manifest
manifest = [(k, s, d) for k, s, d in load_manifest("sdxl-kohya", None) if not d.startswith("int")]
device = torch.device("cuda", 0)
work = pathlib.Path("/dev/shm/mecha_prof"); work.mkdir(parents=True, exist_ok=True)
try:
    B.write_checkpoint(work / "m0.safetensors", manifest, 100, device)
    lps = []
    for i in range(16):
        B.write_lora(work / f"lora{i}.safetensors", manifest, 500 + i, device); lps.append(work / f"lora{i}.safetensors")
    base = amd.model(str(work / "m0.safetensors"))
    recipe = base
    for p in lps:
        recipe = amd.add_difference(recipe, amd.convert(amd.model(str(p)), base), alpha=1.0) | recipe
    def run():
        out = work / "out.safetensors"
        if out.exists(): out.unlink()
        amd.merge(recipe, output=str(out), output_dtype=torch.bfloat16, total_buffer_size=8 << 30)
    run()
    pr = cProfile.Profile(); pr.enable(); run(); pr.disable()
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(40); print(s.getvalue()[:6000])
finally:
    shutil.rmtree(work, ignore_errors=True)
