#!/usr/bin/env python3
"""Aggregates rocprofv3 --pmc counter_collection.csv files per (melf kernel, counter):
prints the mean over dispatches.  Usage: pmc_summary.py DIR [DIR...]"""
import collections
import csv
import glob
import sys

NAMES = [('k_match_mfma', 'k_match_mfma'), ('k_match_gen', 'k_match_gen'), ('k_prep_lplane', 'k_prep_lplane'), ('k_rowsum', 'k_rowsum'),
         ('k_dials', 'k_dials'), ('k_fused_mask', 'k_fused_mask'), ('k_match', 'k_match_dot4'),
         ('k_jpeg_huff', 'k_jpeg_huff'), ('k_jpeg_idct', 'k_jpeg_idct'), ('k_jpeg_color', 'k_jpeg_color')]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[1:]:
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            for (pat, short) in NAMES:
                if pat in r['Kernel_Name']:
                    agg[short][r['Counter_Name']].append(float(r['Counter_Value']))
                    break
for (k, cs) in agg.items():
    for (c, v) in sorted(cs.items()):
        print('%-16s %-28s n=%-4d mean=%.6g' % (k, c, len(v), sum(v) / len(v)))
