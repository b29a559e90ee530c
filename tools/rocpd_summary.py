#!/usr/bin/env python3
"""Summarises rocprofv3 rocpd (.db) outputs into the small text files kept under profiles/.

    python tools/rocpd_summary.py stats gpurun_out/prof_stats/stats_results.db
    python tools/rocpd_summary.py pmc   gpurun_out/prof_fetch/fetch_results.db [more.db ...]
"""
import sqlite3
import sys


def short(name):
    return name.split("(")[0][:60]


def stats(path):
    db = sqlite3.connect(path)
    print(f"# rocprofv3 --kernel-trace --stats   ({path})")
    # rocprofv3 (ROCm 7.2) decodes the kernel descriptor's granulated register count with a granule of 4; gfx950
    # allocates in granules of 8 (MI355X_MICROARCH.md, register files), so the trace's vgpr_count / accum_vgpr_count are
    # HALF the allocation: 60 / 116 / 256 for kernels the compiler reports at 118 / 228 / 512.  Printed doubled =
    # registers allocated per lane, to be read next to profiles/rNN_resource_usage.txt (the compiler's own figures).
    print(f"{'kernel':<28}{'calls':>6}{'total_ms':>12}{'avg_ms':>12}{'min_ms':>12}{'max_ms':>12}{'pct':>8}  grid x wg  vgpr_alloc agpr_alloc sgpr lds")
    rows = db.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration), max(grid_x), max(workgroup_x),"
        " max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_size) from kernels group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows)
    for r in rows:
        print(f"{short(r[0]):<28}{r[1]:>6}{r[2]/1e6:>12.3f}{r[3]/1e6:>12.4f}{r[4]/1e6:>12.4f}{r[5]/1e6:>12.4f}{100*r[2]/total:>8.2f}"
              f"  {r[6]} x {r[7]}  {2 * (r[8] or 0)} {2 * (r[9] or 0)} {r[10]} {r[11]}")


def pmc(paths):
    for path in paths:
        db = sqlite3.connect(path)
        print(f"# rocprofv3 --pmc   ({path})   per-dispatch averages")
        rows = db.execute(
            "select kernel_name, counter_name, count(*), avg(value), avg(duration) from counters_collection "
            "group by kernel_name, counter_name order by kernel_name, counter_name").fetchall()
        print(f"{'kernel':<28}{'counter':<26}{'dispatches':>10}{'avg_value':>20}{'avg_dispatch_ms':>18}")
        for k, c, n, v, d in rows:
            if k.startswith(("void at::", "__amd")):
                continue
            print(f"{short(k):<28}{c:<26}{n:>10}{v:>20.2f}{d/1e6:>18.4f}")


def pmc_json(paths):
    """{kernel: {counter: per-dispatch average}} as JSON (read by bench.py for roofline.traffic)."""
    import json
    out = {}
    for path in paths:
        db = sqlite3.connect(path)
        for k, c, v, d in db.execute("select kernel_name, counter_name, avg(value), avg(duration) from counters_collection "
                                     "group by kernel_name, counter_name"):
            if k.startswith(("void at::", "__amd")):
                continue
            out.setdefault(short(k), {})[c] = v
            out[short(k)].setdefault("dispatch_ms", {})[c] = d / 1e6
        # the shader clock counter of the pass that also holds the VALU activity counter (bench.py's valu_busy needs the two
        # from the same run: profiled clocks differ from pass to pass)
        names = {r[0] for r in db.execute("select distinct counter_name from counters_collection")}
        if "SQ_ACTIVE_INST_VALU" in names and "GRBM_GUI_ACTIVE" in names:
            for k, v in db.execute("select kernel_name, avg(value) from counters_collection where counter_name = 'GRBM_GUI_ACTIVE' "
                                   "group by kernel_name"):
                if not k.startswith(("void at::", "__amd")):
                    out[short(k)]["GRBM_GUI_ACTIVE_same_pass"] = v
    print(json.dumps(out, indent=1, sort_keys=True))


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2])
    elif sys.argv[1] == "pmc-json":
        pmc_json(sys.argv[2:])
    else:
        pmc(sys.argv[2:])
