import sqlite3,sys
sys.path.insert(0,'tools')
from timeline import short
db=sqlite3.connect(sys.argv[1])
rows=db.execute("select name,start,end,stream_id,grid_x,workgroup_x from kernels order by start").fetchall()
idx=[i for i,r in enumerate(rows) if 'k_enc_strings' in r[0]]
i0=idx[-2]; 
t0=rows[i0][1]
for r in rows[i0-1:idx[-1]]:
    print(f"{(r[1]-t0)/1e3:9.1f} {(r[2]-t0)/1e3:9.1f} {(r[2]-r[1])/1e3:8.1f} s{r[3]} {short(r[0])} g={r[4]//max(r[5],1)}x{r[5]}")
