import sqlite3, collections, statistics, sys
db=sqlite3.connect(sys.argv[1])
cols=[r[1] for r in db.execute("pragma table_info(counters_collection)")]
acc=collections.defaultdict(list)
for r in db.execute("select * from counters_collection"):
    d=dict(zip(cols,r))
    if 'scan_kernel' in str(d.get('kernel_name','')):
        acc[d['counter_name']].append(d['value'])
pieces=(1<<30)/1024
print({k:round(statistics.median(v)/pieces,2) for k,v in acc.items()})
