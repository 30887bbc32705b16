#!/usr/bin/env python3
"""Kernel-by-kernel timeline of one encode + decode step from a rocprofv3 kernel trace (rocpd sqlite): durations and the idle gap after every kernel.
usage: python tools/step_gaps.py gpurun_out/prof_x/t_results.db"""
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
rows = cur.execute("select name,start,end from kernels order by start").fetchall()
names = [r[0].split('(')[0].replace('void ', '').replace('mlz::', '') for r in rows]
idx = [i for i, n in enumerate(names) if n.startswith('far_build')]
i0, i1 = idx[-3], idx[-2]
tot = rows[i1][1] - rows[i0][1]
busy = sum(rows[i][2] - rows[i][1] for i in range(i0, i1))
print('step span %.1f us, kernels busy %.1f us, gaps %.1f us, %d kernels' % (tot / 1e3, busy / 1e3, (tot - busy) / 1e3, i1 - i0))
for i in range(i0, i1):
    print('%-44s %8.1f us   gap after %5.2f' % (names[i][:44], (rows[i][2] - rows[i][1]) / 1e3, (rows[i + 1][1] - rows[i][2]) / 1e3))
