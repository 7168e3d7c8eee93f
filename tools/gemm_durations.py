"""per-instance durations of the token GEMM kernels in a rocprofv3 --kernel-trace --stats database (rocpd sqlite):
name<template args>, (grid/256, grid_y, launches, average us, total ms)"""
import re
import sqlite3
import sys
con = sqlite3.connect(sys.argv[1])
cur = con.cursor()
q = ("select name, grid_x/256, grid_y, count(*), avg(end-start)/1000.0, sum(end-start)/1e6 from kernels where name like '%gemm2%' or "
     "name like '%tn_reduce%' or name like '%Cijk%' group by name, grid_x, grid_y order by name")
for r in cur.execute(q):
    m = re.search(r"(gemm2\w*<[^>]*>|tn_reduce\w*|Cijk\w{0,50})", r[0])
    print(m.group(1) if m else r[0][:60], tuple(round(v, 2) if isinstance(v, float) else v for v in r[1:]))
