"""Band-parallel whole-frame Restormer across PROCESSES, end to end through torch.distributed.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29531 tools/strip_e2e.py

--backend gloo (default): every rank uses GPU 0 and the exchange steps are staged through host memory, so the whole path (C ABI band
forward + callbacks + torch.distributed send/recv/all-gather) can be checked on a one-GPU box.  --backend nccl: one GPU per rank, RCCL.
Rank 0 compares the assembled float frame with the one-GPU whole-frame forward and prints one JSON line."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'image-restoration-agent_amd'))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--backend', default='gloo', choices=['gloo', 'nccl'])
    ap.add_argument('--height', type=int, default=200)
    ap.add_argument('--width', type=int, default=264)
    ap.add_argument('--reduced', action='store_true', help='the small test network instead of the 26.1M-parameter denoiser')
    args = ap.parse_args()
    import numpy as np
    import torch
    import torch.distributed as dist
    import torch.nn.functional as F
    from ira_amd.backend import RestormerBackend
    from ira_amd.config import reduced_config, task_config
    from ira_amd.dist import StripParallelRestorer, TorchDistTransport
    from ira_amd.synth import synthetic_noisy_image
    from ira_amd.weights import seeded_state_dict

    rank, world = int(os.environ.get('RANK', '0')), int(os.environ.get('WORLD_SIZE', '1'))
    device = int(os.environ.get('LOCAL_RANK', '0')) if args.backend == 'nccl' else 0
    torch.cuda.set_device(device)
    dev = torch.device('cuda', device)
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    if args.backend == 'nccl':
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group('gloo', rank=rank, world_size=world)
    cfg = reduced_config('BiasFree') if args.reduced else task_config('Real_Denoising')
    be = RestormerBackend(cfg, seeded_state_dict(cfg, 0), device=device)
    h, w = args.height, args.width
    _, noisy = synthetic_noisy_image(h, w, 3, sigma=25.0, seed=3)
    frame = torch.from_numpy(noisy).to(dev)
    rst = StripParallelRestorer(be, rank, world, transport=TorchDistTransport(rank, world, via_host=args.backend == 'gloo'))
    got = rst.restore_frame(frame, return_float=True)
    dist.barrier()
    t0 = time.perf_counter()
    got_u8 = rst.restore_frame(frame)
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    if rank == 0:
        x = frame.permute(2, 0, 1).unsqueeze(0).float().div(255.)
        H, W = ((h + 8) // 8) * 8, ((w + 8) // 8) * 8
        if h % 8 or w % 8:
            x = F.pad(x, (0, W - w if w % 8 else 0, 0, H - h if h % 8 else 0), 'reflect')
        whole = np.transpose(be.forward(x.cpu().numpy())[0], (1, 2, 0))[:h, :w]
        g = got.cpu().numpy()
        d = np.abs(g - whole)
        mse = float(np.mean((g.astype(np.float64) - whole) ** 2))
        rows = d.reshape(h, -1).max(1)
        u8 = be.restore(noisy, tile=None)
        du = np.abs(got_u8.cpu().numpy().astype(int) - u8.astype(int))
        print(json.dumps({'backend': args.backend, 'world': world, 'frame': [h, w], 'psnr_vs_one_gpu_dB': round(10 * np.log10(1.0 / max(mse, 1e-30)), 2),
                          'max_abs': float(d.max()), 'worst_row': int(rows.argmax()), 'worst_row_over_median': round(float(rows.max() / (np.median(rows) + 1e-12)), 2),
                          'u8_max_diff': int(du.max()), 'u8_differing_fraction': round(float((du > 0).mean()), 5), 'seconds_per_frame': round(dt, 4)}), flush=True)
    dist.barrier()
    be.close()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
