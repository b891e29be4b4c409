import sqlite3,sys,glob
f=glob.glob(sys.argv[1]+'/**/*.db',recursive=True)[0]
c=sqlite3.connect(f)
q="select name, count(*), avg(end-start)/1000.0, max(grid_x)/max(workgroup_x), max(lds_size) from kernels group by name order by sum(end-start) desc limit 14"
for r in c.execute(q): print(r[0][:90], [round(x,1) if isinstance(x,float) else x for x in r[1:]])
