#!/usr/bin/env python3
"""Static ISA statistics per kernel: instruction count, scratch (spill / private-array) ops, VMEM / LDS / SMEM ops.

    hipcc --offload-arch=gfx950 -O3 -std=c++20 -S --cuda-device-only -Iinclude -Imycobotgym_amd/csrc \
        mycobotgym_amd/csrc/mcg_hip.hip -o /tmp/mcg.s && python tools/isa_stats.py /tmp/mcg.s
"""
import re, sys
s = open(sys.argv[1]).read()
heads = list(re.finditer(r'^(_Z\w*kernel\w*):', s, re.M))
for k in heads:
    st = k.end(); en = s.index('s_endpgm', st)
    ins = [l for l in s[st:en].split('\n') if l.startswith('\t') and not l.strip().startswith(('.', ';'))]
    sc = [l for l in ins if 'scratch_' in l]
    gen = [l for l in sc if 'Folded' not in l]
    cnt = lambda p: sum(1 for l in ins if re.match(r'\t' + p, l))
    name = re.sub(r'_ZN12_GLOBAL__N_1\d+', '', k.group(1))[:28]
    print(f"{name:30s} instr {len(ins):6d}  scratch {len(sc):5d} (private-array {len(gen):4d})  global {cnt('global_'):4d}"
          f"  ds {cnt('ds_'):5d}  s_load {cnt('s_load'):4d}  v_*f64 {sum(1 for l in ins if '_f64' in l):6d}")
