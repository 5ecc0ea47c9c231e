#!/usr/bin/env python3
"""Guarded loads that the compiler waits for one at a time.

hipcc may not hoist a load over the guard of its own iteration / branch, and when the USE of a guarded load sits inside the same
guard it emits `load; s_waitcnt vmcnt(0); use` per guard: N loads = N memory round trips in a row (round 5 found 12 .. 189 of them
per block of 64 voxels in the seed solver's take, the LASSO certificates' Gram gathers, the SANDI / CZB kernels, every staging
loop).  This lists, per kernel of a device assembly file, the vector loads and how many of them are followed by a full wait before
the next load is issued.

    cd amico_amd/csrc && /opt/rocm/bin/hipcc -DAMX_S2_NW=16 -O3 -std=c++17 --offload-arch=gfx950 --cuda-device-only -S -o /tmp/seed.s amx_seed.hip
    python tools/isa_serial_loads.py /tmp/seed.s

The cure is one of: unconditional loads at clamped indices with the guard on the STORE / select (staging loops); loads under their
guard into registers, arithmetic outside the guard (masked gathers); values parked in LDS once per workgroup (tables)."""
import re, subprocess, sys
lines = open(sys.argv[1]).read().split('\n')
funcs = [(i, m.group(1)) for i, l in enumerate(lines) for m in [re.match(r'^(_Z\S+):', l)] if m]
funcs.append((len(lines), 'end'))
for (a, name), (b, _) in zip(funcs, funcs[1:]):
    seg = lines[a:b]
    loads = [i for i, l in enumerate(seg) if re.search(r'\b(global_load|buffer_load)', l)]
    serial = 0
    for i in loads:
        for j in range(i + 1, min(i + 5, len(seg))):
            if 'vmcnt(0)' in seg[j]:
                serial += 1
                break
            if re.search(r'\b(global_load|buffer_load)', seg[j]):
                break
    if loads:
        dn = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()[:86]
        print('%-88s loads %4d  waited for one by one %4d' % (dn, len(loads), serial))
