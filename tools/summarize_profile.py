"""Condense rocprofv3 output directories (copied back under gpurun_out/) into the tracked files
profiles/<tag>_kernel_stats.csv, profiles/<tag>_summary.md and profiles/traffic.json.

    python tools/summarize_profile.py r01b gpurun_out/prof_r01b_trace gpurun_out/prof_r01b_fetch \
           gpurun_out/prof_r01b_write gpurun_out/prof_r01b_sq [bench.json]

HBM traffic follows /opt/skills/guides/MI355X_MICROARCH.md (HBM / rocprofv3 section): FETCH_SIZE and
WRITE_SIZE are collected in separate --pmc passes, both are in KiB, and on gfx950 FETCH_SIZE
under-reports streaming reads -- "calibrate on a known byte count in your own access pattern".
The calibration kernel is k_snow_threshold of the same run: it reads tc[nt][ncell] once with the
same 8-byte-per-lane coalesced loads as k_run_all, so its true byte count is known exactly.
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load(pattern):
    f = glob.glob(pattern)
    return list(csv.DictReader(open(f[0]))) if f else []


def counters(d):
    rows = load(os.path.join(d, "*", "*counter_collection.csv"))
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        name = r["Kernel_Name"]
        # the timed launches: daily five outputs + totals, plain (not the SUBSET variant of the
        # overlapped inclusive run, which works beside the spin-up's tail)
        key = "run_all" if "k_run_all<false, true, 31u, false>" in name else (
            "snow" if "k_snow_threshold" in name else None)
        if key:
            agg[key][r["Counter_Name"]].append((float(r["Counter_Value"]), int(r["Grid_Size"])))
    return agg


def main():
    tag, d_trace, d_fetch, d_write, d_sq = sys.argv[1:6]
    bench = json.loads(open(sys.argv[6]).read().strip().splitlines()[-1]) if len(sys.argv) > 6 else None
    out = os.path.join(ROOT, "profiles")
    os.makedirs(out, exist_ok=True)
    ks = glob.glob(os.path.join(d_trace, "*", "*kernel_stats.csv"))[0]
    shutil.copy(ks, os.path.join(out, f"{tag}_kernel_stats.csv"))
    stats = list(csv.DictReader(open(ks)))
    nt, W, rows_blk = 365, 4096, 512
    cells_launch = rows_blk * W
    cells_pmc = 1024 * W  # the PMC passes run bench.py --rows 1024 (same launch shape, 2 launches)
    f, w, q = counters(d_fetch), counters(d_write), counters(d_sq)
    mean = lambda xs: sum(v for v, _ in xs) / len(xs)
    # calibration on k_snow_threshold (known bytes: one 8-byte read per cell-day)
    snow_fetch = mean(f["snow"]["FETCH_SIZE"]) * 1024.0
    snow_true = float(cells_pmc) * nt * 8
    factor = snow_fetch / snow_true
    ra_fetch_raw = mean(f["run_all"]["FETCH_SIZE"]) * 1024.0
    ra_write = mean(w["run_all"]["WRITE_SIZE"]) * 1024.0
    ra_fetch = ra_fetch_raw / factor
    alg_read, alg_write = cells_launch * nt * 24.0, cells_launch * nt * 40.0
    traffic = ra_fetch + ra_write
    sq = {k: mean(v) for k, v in q["run_all"].items()}
    waves = sq.get("SQ_WAVES", 0)
    md = [f"# rocprofv3 summary `{tag}` (MI355X, ROCm 7.2)", ""]
    md.append("Commands (each its own run; counters never combined with trace domains other than "
              "`--kernel-trace`):")
    md.append("```")
    md.append("rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py")
    md.append("rocprofv3 --pmc FETCH_SIZE --kernel-trace ... -- python3 bench.py --rows 1024 --steps 1 --warmup 0 --cpu-seconds 0")
    md.append("rocprofv3 --pmc WRITE_SIZE --kernel-trace ... (same)")
    md.append("rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU "
              "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU SQ_WAVES --kernel-trace ... (same)")
    md.append("```")
    md += ["", "## Kernel time (`--kernel-trace --stats`, default `bench.py`: 4096x4096x365, 3+1 steps)", "",
           "| kernel | calls | avg ms | % |", "|---|---|---|---|"]
    for r in stats[:9]:
        md.append(f"| `{r['Name'][:60]}` | {r['Calls']} | {float(r['AverageNs'])/1e6:.3f} | {float(r['Percentage']):.2f} |")
    ra = next(r for r in stats if "k_run_all<false, true, 31u, false>" in r["Name"])
    avg_ms = float(ra["AverageNs"]) / 1e6
    md += ["", f"Dominant kernel of the timed region: `k_run_all<false,true>` — {avg_ms:.2f} ms per launch of "
           f"{rows_blk}x{W} cells x {nt} days = {cells_launch*nt:.3e} cell-days "
           f"(algorithmic {64*cells_launch*nt/1e9:.1f} GB => {64*cells_launch*nt/avg_ms/1e6:.1f} GB/s, "
           f"{64*cells_launch*nt/avg_ms/1e6/8000:.4f} of 8 TB/s)."]
    if bench:
        md.append(f"`bench.py` of the same run measured {bench['roofline']['avg_launch_ms']:.2f} ms per launch with "
                  f"stream events (value {bench['value']:.4e} cell-days/s).")
    md += ["", "## HBM traffic of one `k_run_all` launch (PMC, separate passes)", "",
           f"* calibration: `k_snow_threshold` FETCH_SIZE = {snow_fetch/1e9:.3f} GB for {snow_true/1e9:.3f} GB "
           f"truly read => FETCH_SIZE reports x{factor:.3f} of this access pattern (8 B/lane coalesced); "
           "the guide's factor for wide streaming reads is 0.5",
           f"* FETCH_SIZE raw {ra_fetch_raw/1e9:.2f} GB -> corrected **{ra_fetch/1e9:.2f} GB** "
           f"(algorithmic reads {alg_read/1e9:.2f} GB)",
           f"* WRITE_SIZE **{ra_write/1e9:.2f} GB** (algorithmic writes {alg_write/1e9:.2f} GB; exact for streaming stores)",
           f"* total **{traffic/1e9:.2f} GB** vs algorithmic {(alg_read+alg_write)/1e9:.2f} GB: "
           f"x{traffic/(alg_read+alg_write):.3f} -- no wasted re-reads; the kernel is not HBM-bound"]
    if sq:
        per_wave_day = sq["SQ_INSTS_VALU"] / waves / nt
        md += ["", "## SQ counters of one `k_run_all` launch", "",
               f"* waves {waves:.0f}; VALU instructions per wave-day {per_wave_day:.0f}; SALU per wave-day "
               f"{sq['SQ_INSTS_SALU']/waves/nt:.0f}",
               f"* SQ_WAVE_CYCLES {sq['SQ_WAVE_CYCLES']:.3e}, SQ_ACTIVE_INST_VALU {sq['SQ_ACTIVE_INST_VALU']:.3e} "
               f"({sq['SQ_ACTIVE_INST_VALU']/sq['SQ_WAVE_CYCLES']:.2f} of wave cycles issuing VALU), "
               f"SQ_WAIT_ANY {sq['SQ_WAIT_ANY']/sq['SQ_WAVE_CYCLES']:.2f}, "
               f"SQ_WAIT_INST_ANY {sq['SQ_WAIT_INST_ANY']/sq['SQ_WAVE_CYCLES']:.2f} of wave cycles",
               "* reading: the launch is FP64-VALU-issue bound (pointwise transcendental stream), "
               f"as SURVEY.md 8(d) predicted; HBM runs at {64*cells_launch*nt/avg_ms/1e6/8000:.0%} of its peak rate"]
    open(os.path.join(out, f"{tag}_summary.md"), "w").write("\n".join(md) + "\n")
    valu = {}
    if sq:
        # 2 waves per SIMD: the vector pipe is busy when either wave issues a VALU instruction
        valu = {"valu_busy_frac": 2.0 * sq["SQ_ACTIVE_INST_VALU"] / sq["SQ_WAVE_CYCLES"],
                "valu_insts_per_wave_day": sq["SQ_INSTS_VALU"] / waves / nt,
                "salu_insts_per_wave_day": sq["SQ_INSTS_SALU"] / waves / nt}
    tj = {f"{rows_blk}x{W}x{nt}": {**valu, "hbm_bytes_per_launch": traffic, "fetch_bytes_corrected": ra_fetch,
                                    "write_bytes": ra_write, "fetch_size_calibration": factor,
                                    "algorithmic_bytes": alg_read + alg_write, "source": f"profiles/{tag}_summary.md",
                                    # the kernel sources these counters were measured on (bench.py prints the hash of
                                    # the sources it runs; a mismatch shows in every later bench line)
                                    "kernel_sources_sha16": (bench or {}).get("roofline", {}).get("kernel_sources_sha16")}}
    json.dump(tj, open(os.path.join(out, "traffic.json"), "w"), indent=1)
    if bench:
        json.dump(bench, open(os.path.join(out, f"{tag}_bench.json"), "w"), indent=1)
    print("\n".join(md))


if __name__ == "__main__":
    main()
