import sys, time
sys.path.insert(0, '/root/repo')
import torch
from new_cg_variants_amd import problems as P
from new_cg_variants_amd.device import DeviceCSR
for name in ('s1', 's2', 's3', 's4b', 's2_8th'):
    A = P.WORKLOADS[name]['make']()
    for kn in ({}, {'PRCG_VALDICT': '0'}):
        t0 = time.perf_counter(); op = DeviceCSR(A, knobs=kn); dt = time.perf_counter() - t0
        s = op.schedule(); lay = op.layout()
        print(f'{name} {kn}: set-up {dt:.2f} s  pattern {s["pattern"]} geometry {lay["geometry"]} sweep waves {lay["sweep_waves"]}', flush=True)
        op.close()
