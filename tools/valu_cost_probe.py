"""Issue cost of single vector instructions on gfx950 (tools/experiments/valu_cost_probe.hip): cycles per wave-instruction and SIMD with one
and two waves per SIMD.   python tools/valu_cost_probe.py   (GPU box; builds build/valu_cost_probe.so if missing)"""
import ctypes
import json
import os
import subprocess

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "build", "valu_cost_probe.so")
SRC = os.path.join(ROOT, "tools", "experiments", "valu_cost_probe.hip")
OPS = {0: "v_exp_f32", 1: "v_exp_f16", 2: "v_fma_f32", 3: "v_pk_fma_f32", 4: "v_cvt_pk_f16_f32", 5: "v_max3_f32", 6: "v_rcp_f32",
       7: "v_exp_f16_sdwa (high half)", 8: "v_pk_mul_f32", 9: "v_mov_b32", 10: "v_pk_add_f16"}

if not os.path.exists(SO) or os.path.getmtime(SO) < os.path.getmtime(SRC):
    os.makedirs(os.path.dirname(SO), exist_ok=True)
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", SRC, "-o", SO])
L = ctypes.CDLL(SO)
L.valu_cost_probe.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
out = torch.zeros(256 * 512, device="cuda")
cyc = torch.zeros(256 * 8, dtype=torch.int64, device="cuda")
reps = 200
res = {}
for op, name in OPS.items():
    row = {}
    for waves in (4, 8):       # one / two waves per SIMD (one workgroup per CU: 256 workgroups)
        for _ in range(2):
            assert L.valu_cost_probe(op, waves, out.data_ptr(), cyc.data_ptr(), reps, 0) == 0
        torch.cuda.synchronize()
        c = cyc[:256 * waves].float().mean().item()
        # s_memtime ticks at 100 MHz on this chip family; convert with the wall time instead: time the launch
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            L.valu_cost_probe(op, waves, out.data_ptr(), cyc.data_ptr(), reps, 0)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 5
        n_inst = reps * 64 * (waves // 4)          # wave-instructions per SIMD
        row["%d wave(s)/SIMD" % (waves // 4)] = {"us": round(us, 1), "ns_per_inst_per_simd": round(us * 1e3 / n_inst, 2), "counter_per_inst": round(c / (reps * 64), 2)}
    res[name] = row
print(json.dumps(res, indent=1))
