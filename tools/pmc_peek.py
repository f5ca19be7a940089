import sqlite3, glob
for db in sorted(glob.glob("gpurun_out/pmc/*/p_results.db")):
    c = sqlite3.connect(db)
    print(db, c.execute("select counter_name, count(*), sum(counter_value) from pmc_events group by counter_name").fetchall())
