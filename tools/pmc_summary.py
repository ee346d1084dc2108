#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc counter_collection.csv per kernel (mean per dispatch)."""
import csv, os, re, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
KEEP_TEMPLATES = os.environ.get("PMC_KEEP_TEMPLATES") == "1"  # tell the instantiations of a kernel template apart (count / emit pass, digit widths)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    name = r['Kernel_Name'].replace('gs::(anonymous namespace)::', '').replace('rocprim::ROCPRIM_400200_NS::detail::', 'rp::')
    name = re.sub(r'\(.*', '', name)[:70] if KEEP_TEMPLATES else re.sub(r'<.*', '', name)[:40]
    acc[name][r['Counter_Name']].append(float(r['Counter_Value']))
for name, cs in sorted(acc.items()):
    print(name, {c: round(sum(v) / len(v), 1) for c, v in cs.items()}, 'dispatches', len(next(iter(cs.values()))))
