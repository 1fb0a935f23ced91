"""Accuracy of plain fp16 weights (no lo plane) in the 1x1 layers of Restormer levels 1-2 (dim 48 / 96: the levels of the fused kernels),
measured ON THE GPU through the harness: for each ablation set, |PSNR(build, clean) - PSNR(oracle, clean)| on the reference-generated
harness goldens, the 18 robustness-sweep goldens and three 512^2 tiles of the benchmark frame (oracle on the CPU).  Zeroing the lo planes
in the blob (IRA_ABLATE_LO_L12, ira_amd/pack.py) is numerically what a kernel that skips the lo-plane MFMAs computes.
usage: python tools/ablate_lo_restormer.py [sets ...]   e.g.  '' qkv pin qkv,pin qkv,pin,pout   (tags also as kind@stage, e.g. pin@decoder_level1)
environment: ABLATE_UNSEEN=1 adds eleven cases on tasks / weight seeds / sizes no golden was made from (oracle on the CPU);
ABLATE_RULES=r1,r2,... runs the first set under each IRA_FFN_LO_STAGES rule (eight digits, csrc/ira_rt_restormer.hip::ffn_lo_rule);
ABLATE_PLANES=2 creates the handles with every weight as fp16 hi + lo (the noise floor of the statistic).
Results of round 3: profiles/r03_restormer_lo_plane_ablation.txt, DESIGN.md 4."""
import glob, json, os, sys
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, os.path.join(R, 'image-restoration-agent_amd')); sys.path.insert(0, R)
import numpy as np, torch
torch.set_num_threads(16)
from ira_amd.backend import RestormerBackend
from ira_amd.config import RestormerConfig, task_config
from ira_amd.weights import seeded_state_dict, to_torch
from oracle import restormer_oracle as O

PLANES = int(os.environ.get('ABLATE_PLANES', '1'))      # 2: every MFMA weight as fp16 hi + lo (ira_set_weight_planes)
GOLD = os.path.join(R, 'tests', 'golden')
files = sorted(glob.glob(os.path.join(GOLD, 'harness_*.npz'))) + sorted(glob.glob(os.path.join(GOLD, 'sweep_*.npz')))
sets = sys.argv[1:] or ['', 'qkv', 'pin', 'pout', 'qkv,pin', 'qkv,pin,pout']
# benchmark tiles: oracle once
cfgB = task_config('Real_Denoising'); sdB = seeded_state_dict(cfgB, 0)
clean, noisy = O.synthetic_noisy_image(2160, 3840, 3, sigma=25.0, seed=0)
tiles = [(0, 0), (960, 1440), (1648, 3328)]
refB = []
for (y0, x0) in tiles:
    refB.append(O.restore_image(noisy[y0:y0 + 512, x0:x0 + 512], to_torch(sdB), cfgB.to_dict()))
    print('oracle tile', y0, x0, 'done', flush=True)
# cases on weights and images no golden was made from (ABLATE_UNSEEN=1): the agent-API tests' three, and eight more tasks / seeds / sizes
unseen = []
if os.environ.get('ABLATE_UNSEEN'):
    specs = [('Motion_Deblurring', 7, 72, 88, 3, 11, None, 32), ('Real_Denoising', 7, 64, 80, 3, 12, 48, 16), ('Gaussian_Gray_Denoising', 7, 56, 56, 1, 13, None, 32),
             ('Motion_Deblurring', 3, 96, 120, 3, 31, None, 32), ('Deraining', 5, 80, 112, 3, 32, None, 32), ('Real_Denoising', 9, 128, 128, 3, 33, None, 32),
             ('Single_Image_Defocus_Deblurring', 4, 72, 104, 3, 34, None, 32), ('Gaussian_Color_Denoising', 6, 88, 88, 3, 35, None, 32),
             ('Real_Denoising', 2, 160, 160, 3, 36, 96, 32), ('Motion_Deblurring', 8, 64, 64, 3, 37, None, 32), ('Deraining', 1, 120, 88, 3, 38, 64, 16)]
    for (task, wseed, hh, ww, ch, iseed, tile, ov) in specs:
        cfgU = task_config(task); sdU = seeded_state_dict(cfgU, wseed)
        cl, nz = O.synthetic_noisy_image(hh, ww, ch, seed=iseed)
        rf = O.restore_image(nz, to_torch(sdU), cfgU.to_dict(), tile=tile, tile_overlap=ov)
        unseen.append((f'unseen {task} w{wseed} {hh}x{ww}', cfgU, sdU, cl, nz, rf, tile, ov))
        print('oracle', unseen[-1][0], 'done', flush=True)
rules = [r for r in os.environ.get('ABLATE_RULES', '').split(',') if r] or [None]
if len(rules) > 1 or rules[0]: sets = [(st, r) for st in sets[:1] for r in rules]
else: sets = [(st, None) for st in sets]
for st, rule in sets:
    os.environ['IRA_ABLATE_LO_L12'] = st
    if rule: os.environ['IRA_FFN_LO_STAGES'] = rule
    worst = 0.0; rows = []
    for (name, cfgU, sdU, cl, nz, rf, tile, ov) in unseen:
        with RestormerBackend(cfgU, sdU, device=0, weight_planes=PLANES) as be:
            out = be.restore(nz, tile=tile, tile_overlap=ov)
        d = O.psnr_u8(out, cl) - O.psnr_u8(rf, cl)
        rows.append((name, d, int(np.abs(out.astype(int) - rf.astype(int)).max())))
        worst = max(worst, abs(d))
    for f in files:
        z = np.load(f, allow_pickle=False)
        cfg = RestormerConfig(**json.loads(str(z['cfg'])))
        kw = {}
        if 'residual_scale' in z.files: kw['residual_scale'] = float(z['residual_scale'])
        sd = seeded_state_dict(cfg, int(z['seed']), **kw)
        with RestormerBackend(cfg, sd, device=0, weight_planes=PLANES) as be:
            tile = int(z['tile']) if int(z['tile']) > 0 else None
            out = be.restore(z['noisy'], tile=tile, tile_overlap=int(z['overlap']))
        ref = z['restored']; cl = z['clean']
        d = O.psnr_u8(out, cl) - O.psnr_u8(ref, cl)
        rows.append((os.path.basename(f), d, int(np.abs(out.astype(int) - ref.astype(int)).max())))
        worst = max(worst, abs(d))
    with RestormerBackend(cfgB, sdB, device=0, weight_planes=PLANES) as be:
        for (y0, x0), rf in zip(tiles, refB):
            out = be.restore(noisy[y0:y0 + 512, x0:x0 + 512])
            cl = clean[y0:y0 + 512, x0:x0 + 512]
            d = O.psnr_u8(out, cl) - O.psnr_u8(rf, cl)
            rows.append((f'bench tile {y0},{x0}', d, int(np.abs(out.astype(int) - rf.astype(int)).max())))
            worst = max(worst, abs(d))
    if rule: st = f'{st} rule {rule}'
    if PLANES != 1: st = f'{st} weight_planes {PLANES}'
    if unseen: print('   unseen cases:', [(n, f'{d:+.2e}') for (n, d, _) in rows[:len(unseen)]], flush=True)
    print(f"ablate [{st}]: worst |dPSNR| {worst:.2e} dB over {len(rows)} cases; the five worst:", sorted(rows, key=lambda r: -abs(r[1]))[:5], flush=True)
