#!/usr/bin/env python3
"""A/B of library switches on resident NODDI fits of several sizes, one process, one data set (round 6: the left-over fork).
   python tools/r06/fork_ab.py "50000 100000 200000 300000 1000000" "AMX_FORK=0" "AMX_FORK=2" "AMX_FORK=1" "AMX_FORK=3 AMX_FORK_PRIO=1"
Every configuration gets a fresh context (the library reads its switches at amx_ctx_create); maps are compared with the first one's."""
import os
import sys
import time
import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..', '..'))
sys.path.insert(0, ROOT)


def main():
    import torch
    from amico_amd import _capi, synthetic as S
    sizes = [int(v) for v in sys.argv[1].split()]
    configs = sys.argv[2:] or ['AMX_FORK=0']
    steps = int(os.environ.get('AB_STEPS', '10'))
    nmax = max(sizes)
    dirs = S.fibonacci_hemisphere(500)
    ht = S.build_htable(dirs)
    sch = S.make_scheme(seed=0)
    K = S.noddi_kernels(sch, dirs)
    snr = float(os.environ.get('AB_SNR', '0'))
    y_h, d_h = S.noddi_signals(nmax, K, ht, sch, seed=1, **({'snr': snr} if snr > 0 else {}))
    dev = torch.device('cuda', 0)
    y = torch.from_numpy(y_h).to(dev); d = torch.from_numpy(d_h).to(dev)
    L = _capi.lib()
    ref = {}
    for cfg in configs:
        kv = dict(t.split('=', 1) for t in cfg.split())
        for k, v in kv.items():
            os.environ[k] = v
        ctx = _capi.Context(0)
        for k in kv:
            del os.environ[k]
        lut = _capi.upload_noddi(ctx, K, ht, sch.dwi_idx, False)
        ctx.set_profiling(os.environ.get('AB_PROFILING', '1') != '0')
        stream = torch.cuda.current_stream().cuda_stream
        for n in sizes:
            est = torch.zeros((n, 3), dtype=torch.float64, device=dev)

            def fit():
                ctx.check(L.amx_noddi_fit_device(ctx._h, lut._h, y.data_ptr(), d.data_ptr(), n, 0.5, 1e-3, 0, est.data_ptr(), None, None, None, stream))
            for _ in range(3):
                fit(); ctx.sync(stream)
            torch.cuda.synchronize()
            ts = []
            for _ in range(steps):
                t0 = time.perf_counter(); fit(); ctx.sync(stream); ts.append(time.perf_counter() - t0)
            ms = 1e3 * float(np.median(ts))
            e = est.cpu().numpy()
            if n not in ref:
                ref[n] = e
                cmp_ = ''
            else:
                df = np.abs(e - ref[n]).max(axis=1)
                cmp_ = ' | vs first: %d voxels differ, max %.1e' % (int((df > 0).sum()), float(df.max()))
            ss = ctx.last_seed_stats()
            print('%-40s %8d voxels: %7.3f ms (min %.3f)  %7.2f M voxels/s | left %d %d %d%s' % (
                cfg, n, ms, 1e3 * min(ts), n / ms / 1e3, ss.get('leftover_stage1', -1), ss.get('leftover_lasso', -1), ss.get('leftover_stage3', -1), cmp_), flush=True)
            del est
        lut.close(); ctx.close()


if __name__ == '__main__':
    main()
