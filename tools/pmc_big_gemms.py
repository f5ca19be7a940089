import sqlite3, sys
c=sqlite3.connect(sys.argv[1])
q="select K.dispatch_id, P.name, P.counter_value, P.duration, K.grid_size_x,K.grid_size_y,K.grid_size_z, K.workgroup_size_x from (select * from pmc_events) P join rocpd_kernel_dispatch K on K.dispatch_id=P.dispatch_id order by K.dispatch_id"
rows=[r for r in c.execute(q).fetchall() if 'gemm_kernel' in r[1]]
half=rows[len(rows)//2:]
for r in half:
    if r[2]*2048>400e6: print("%7.1f MB fetch(x2) %7.1f us grid=(%d,%d,%d) %s"%(r[2]*2048/1e6,r[3]/1e3,r[4]//r[7],r[5],r[6],r[1][13:60]))
