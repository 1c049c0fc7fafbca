#!/usr/bin/env python3
"""Attribute a kernel's instructions to source lines (needs an ISA dump compiled with -gline-tables-only).
   usage: isa_lines.py file.s kernel_substring [opcode_substring]"""
import re, sys, collections
s = open(sys.argv[1]).read(); want = sys.argv[2]; opf = sys.argv[3] if len(sys.argv) > 3 else None
files = {}
for m in re.finditer(r'\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', s):
    files[int(m.group(1))] = (m.group(3) or m.group(2)).split('/')[-1]
k = [m for m in re.finditer(r'^(_Z\w*):', s, re.M) if want in m.group(1)][0]
st = k.end(); en = s.index('s_endpgm', st)
cur = None; cnt = collections.Counter()
for l in s[st:en].split('\n'):
    t = l.strip()
    if t.startswith('.loc'):
        p = t.split(); cur = (files.get(int(p[1]), p[1]), int(p[2])); continue
    if l.startswith('\t') and t and not t.startswith(('.', ';')):
        if opf is None or opf in t.split()[0] or (opf == 'private' and 'scratch_' in t and 'Folded' not in t):
            cnt[cur] += 1
print(sum(cnt.values()))
for key, c in cnt.most_common(40): print(key, c)
