"""Per-stage timing of the agent tool chain (dev aid for bench.py --workload agent_chain)."""
import os, sys, time, tempfile
from pathlib import Path
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'image-restoration-agent_amd')); sys.path.insert(0, ROOT)
import numpy as np
from PIL import Image
from ira_amd import tools as T
from ira_amd.synth import synthetic_noisy_image
from ira_amd.swinir import seeded_state_dict as sw_seeded, task_config as sw_task
cache = T.BackendCache(device=0, state_dict_loader=T.seeded_state_dict_loader(0))
swc = T.SwinIRCache(device=0, state_dict_loader=lambda task, name, wd: sw_seeded(sw_task(task, 4, True), 0))
tools = [T.Restormer('denoising', cache=cache), T.Restormer('motion_deblurring', cache=cache), T.SwinIR('super_resolution', 'psnr', cache=swc)]
_, noisy = synthetic_noisy_image(540, 960, 3, sigma=15.0, seed=0)
root = Path(tempfile.mkdtemp())
for rep in range(3):
    d = root / f'r{rep}' / 'in'; d.mkdir(parents=True)
    Image.fromarray(noisy).save(d / 'input.png')
    cur = d
    for i, tool in enumerate(tools):
        out = root / f'r{rep}' / f'o{i}'; out.mkdir()
        t0 = time.perf_counter()
        img = T.read_image(sorted(cur.glob('*'))[0])
        t1 = time.perf_counter()
        be = cache.get(tool.opt_task) if i < 2 else swc.get(tool.opt_task, tool.model_name)
        t2 = time.perf_counter()
        res = be.restore(img)
        t3 = time.perf_counter()
        T.write_png(out / 'output.png', res)
        t4 = time.perf_counter()
        print(f"rep {rep} {tool.tool_name:12s} read {1e3*(t1-t0):7.1f} ms  get-backend {1e3*(t2-t1):7.1f}  restore {1e3*(t3-t2):7.1f}  write {1e3*(t4-t3):7.1f}  out {res.shape}")
        cur = out
