"""Print per-kernel PMC counter averages from a rocprofv3 results.db.  usage: python scripts/pmc_dump.py <results.db>"""
import sqlite3, sys, re
db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
pmc = [t for t in tabs if t.startswith("rocpd_pmc_event")][0]
info = [t for t in tabs if t.startswith("rocpd_info_pmc")][0]
disp = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
q = ("select s.kernel_name, i.name, count(*), avg(e.value) from %s e join %s i on e.pmc_id = i.id "
     "join %s d on e.event_id = d.event_id join %s s on d.kernel_id = s.id group by 1, 2 order by 1, 2" % (pmc, info, disp, sym))
for name, cname, n, v in db.execute(q):
    if "hupr" in name:
        print("%-60s %-28s n=%-3d avg=%.4g" % (re.sub(r"\(.*", "", name)[:60], cname, n, v))
