#!/usr/bin/env python3
"""Summarise `hipcc -Rpass-analysis=kernel-resource-usage` output: VGPRs, scratch (spills), occupancy per kernel.
usage: hipcc ... -c file.hip -Rpass-analysis=kernel-resource-usage 2> res.txt; tools/kernel_resources.py res.txt [filter...]"""
import re
import subprocess
import sys

txt = open(sys.argv[1]).read()
filters = sys.argv[2:]
blocks = re.split(r'remark: [^\n]*Function Name: ', txt)[1:]
names = [b.split('\n')[0].strip() for b in blocks]
dem = subprocess.run(['c++filt'], input='\n'.join(names), capture_output=True, text=True).stdout.split('\n')
for b, d in zip(blocks, dem):
    def g(k):
        m = re.search(k + r': (\d+)', b)
        return int(m.group(1)) if m else -1
    d = d.replace('fpm::', '').replace('FFTPlan', 'P')
    d = re.sub(r'\(.*', '', d)
    d = d.replace('void ', '')
    scratch = g(r'ScratchSize \[bytes/lane\]')
    if filters and not any(f in d for f in filters) and scratch <= 0:
        continue
    print("%-88s vgpr %3d agpr %3d scratch %4d occ %d" % (d[:88], g('VGPRs'), g('AGPRs'), scratch, g(r'Occupancy \[waves/SIMD\]')))
