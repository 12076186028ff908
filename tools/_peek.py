import sqlite3, sys
con = sqlite3.connect(sys.argv[1])
print([r[1] for r in con.execute("pragma table_info(regions)")])
for r in con.execute("select * from regions limit 3"): print(r)
