import sqlite3,sys
con=sqlite3.connect(sys.argv[1]); cur=con.cursor()
q="select name, grid_x/256, grid_y, count(*), avg(end-start)/1000.0, sum(end-start)/1e6 from kernels where name like '%gemm2%' or name like '%tn_reduce%' or name like '%Cijk%' group by name, grid_x, grid_y order by name"
for r in cur.execute(q): print(r[0].split('(')[0][-40:] if 'Cijk' not in r[0] else r[0][:60], r[1:])
