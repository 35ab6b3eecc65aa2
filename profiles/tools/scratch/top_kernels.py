import sqlite3, glob, sys
db = sqlite3.connect(glob.glob(sys.argv[1])[0])
for r in db.cursor().execute("select * from top_kernels limit 9"):
    print([(x[:60] if isinstance(x, str) else x) for x in r])
