#!/usr/bin/env python
"""Summaries of a tools/profile_round.sh session (rocprofv3 CSV output) in the text / JSON forms kept under profiles/:

    python tools/profile_collect.py gpurun_out/<tag> <tag>      ->  gpurun_out/<tag>/summary/<tag>_<config>_{kernel_stats,pmc}.txt,
                                                                     <tag>_pmc_calibration.txt, traffic.json, <tag>_configs.jsonl

Read traffic is taken from the request counter itself (TCC_EA0_RDREQ_sum x 128 B: tools/pmc_reqsize_cal.sh shows, in the same session
and on a kernel whose bytes are known exactly, that every L2 read request on gfx950 is a 128-byte line), written traffic from WRITE_SIZE."""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

# C3: zz_local_trackw_kernel (one proposal per lane) or zz_local_track_kernel (8-lane groups) -- matched by their common prefix
MAIN = {"C3": "zz_local_track", "C3X": "zz_local_spec8_kernel", "C2": "bps_run_kernel", "C4": "zz_logistic_lds_kernel",
        "C4T": "zz_logistic_lds_kernel", "C5": "zz_general_run_kernel",
        # config C3G (graphs that are not the 2-d lattice): tracked / moving evaluation on the 3-d lattice and on a random pattern
        "C3G": "zz_local_trackp_kernel", "C3GX": "zz_local_spec8g_kernel", "C3G_random6": "zz_local_trackp_kernel",
        "C3GX_random6": "zz_local_spec8g_kernel", "C3G_random8": "zz_local_trackp_kernel", "C3GX_random8": "zz_local_spec8g_kernel",
        # round 5: a rank's share of the 4096-chain ensemble on 2 / 4 / 8 GPUs (the strong-scaling proxy) and the 256 x 256 lattice
        "C3_w2048": "zz_local_trackp_kernel", "C3_w1024": "zz_local_trackp2_kernel", "C3_w512": "zz_local_trackp2_kernel",
        "C3X_w1024": "zz_local_spec8_kernel", "C3X_w512": "zz_local_spec8_kernel", "C3_g256": "zz_local_trackp_big_kernel",
        # round 6: the line layout (opt-in form, PDMP_TRACK_LINES=1) and the lattice with the flow's refresh clock on (bench.py --lambda-ref 1)
        "C3L": "zz_local_trackl_kernel", "C3R": "zz_local_spec8_kernel"}
CMD = {"C3X": "C3 --exact", "C4T": "C4 --tracked", "C3G": "C3G", "C3GX": "C3G --exact", "C3G_random6": "C3G --graph random6",
       "C3GX_random6": "C3G --graph random6 --exact", "C3G_random8": "C3G --graph random8", "C3GX_random8": "C3G --graph random8 --exact",
       "C3_w2048": "C3 --chains 2048", "C3_w1024": "C3 --chains 1024", "C3_w512": "C3 --chains 512", "C3X_w1024": "C3 --exact --chains 1024",
       "C3X_w512": "C3 --exact --chains 512", "C3_g256": "C3 --grid 256 --chains 1024", "C3L": "C3 (PDMP_TRACK_LINES=1)", "C3R": "C3 --lambda-ref 1.0"}


def rows(path):
    return list(csv.DictReader(open(path))) if os.path.exists(path) else []


def find(out, sub, name):
    g = glob.glob(os.path.join(out, sub, "**", name), recursive=True)
    return g[0] if g else os.path.join(out, sub, name)


def counter_means(path, kernel_substr, last=None):
    acc = collections.defaultdict(list)
    for r in rows(path):
        if kernel_substr in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: (sum(v[-last:] if last else v) / len(v[-last:] if last else v), len(v)) for k, v in acc.items()}


def main():
    out, tag = sys.argv[1], sys.argv[2]
    summ = os.path.join(out, "summary")
    os.makedirs(summ, exist_ok=True)
    # ---- calibration (tools/pmc_reqsize_cal.sh): what the fabric-side request counters count on gfx950
    cal_lines = ["# PMC calibration on pdmp::sector_probe_kernel (tools/pmc_reqsize_cal.sh, separate --pmc passes): known useful bytes per launch",
                 "# against the L2 <-> fabric request counters.  Findings on gfx950: EVERY read request is a 128-byte line (TCC_EA0_RDREQ_128B ==",
                 "# TCC_EA0_RDREQ; the 32B/64B classes stay 0) -- a lane that reads one 32-byte sector pulls its whole 128-byte line from HBM --,",
                 "# so FETCH_SIZE's formula (requests x 64 B) is 2x low for line-sized requests and 2x high against the useful bytes of sector",
                 "# reads; writes leave L2 as 32-byte (or 64-byte) partial writes, WRITE_SIZE counts them as they are.",
                 "# HBM bytes read = 128 x TCC_EA0_RDREQ_sum; HBM bytes written = 1024 x WRITE_SIZE = 32 x (WRREQ - WRREQ_64B) + 64 x WRREQ_64B."]
    names = {0: "32-byte sector per lane, read", 2: "64-byte record per lane, read", 7: "128-byte line per 4 lanes, read",
             1: "32-byte sector per lane, read + write-back", 3: "64-byte record per lane, read + write-back", 8: "128-byte line per 4 lanes, read + write-back"}
    cp = os.path.join(out, "cal.jsonl")
    if os.path.exists(cp):
        for ln in open(cp):
            if not ln.startswith("{"):
                continue
            k = json.loads(ln)
            rd = 128 * k.get("TCC_EA0_RDREQ_sum", 0.0)
            wr = 32 * (k.get("TCC_EA0_WRREQ_sum", 0.0) - k.get("TCC_EA0_WRREQ_64B_sum", 0.0)) + 64 * k.get("TCC_EA0_WRREQ_64B_sum", 0.0)
            cal_lines.append(f"mode {k['mode']} ({names.get(k['mode'], '?')}): useful read {k['known_read']:.4g} B, written {k['known_written']:.4g} B per launch; "
                             f"RDREQ {k.get('TCC_EA0_RDREQ_sum', 0):.4g} (128B {k.get('TCC_EA0_RDREQ_128B_sum', 0):.4g}, 64B {k.get('TCC_EA0_RDREQ_64B_sum', 0):.3g}, "
                             f"32B {k.get('TCC_EA0_RDREQ_32B_sum', 0):.3g}) -> HBM read {rd:.4g} B = {rd / max(k['known_read'], 1):.2f} x useful; "
                             f"WRREQ {k.get('TCC_EA0_WRREQ_sum', 0):.4g} (64B {k.get('TCC_EA0_WRREQ_64B_sum', 0):.4g}) -> HBM written {wr:.4g} B"
                             + (f" = {wr / k['known_written']:.2f} x useful" if k["known_written"] else ""))
    open(os.path.join(summ, f"{tag}_pmc_calibration.txt"), "w").write("\n".join(cal_lines) + "\n")
    # ---- per configuration
    import bench
    traffic = {"source_hash": bench.source_hash(), "round_tag": tag,
               "accounting": "HBM bytes read = 128 x TCC_EA0_RDREQ_sum (every L2 read request is a 128-byte line on gfx950), written = 1024 x WRITE_SIZE",
               "configs": {}}
    lines = []
    for C, kern in MAIN.items():
        bj = os.path.join(out, f"{C}_bench.json")
        if not os.path.exists(bj) or os.path.getsize(bj) == 0:
            continue
        B = json.loads(open(bj).read().strip().splitlines()[-1])
        lines.append(json.dumps(B))
        ks = rows(find(out, f"{C}_stats", "st_kernel_stats.csv"))
        kt = rows(find(out, f"{C}_stats", "st_kernel_trace.csv"))
        with open(os.path.join(summ, f"{tag}_{C}_kernel_stats.txt"), "w") as f:
            f.write(f"# rocprofv3 --kernel-trace --stats of: python bench.py --config {CMD.get(C, C)} --steps {B['steps']} --warmup {B['warmup']} "
                    f"--no-cpu-baseline --ess-batches 0\n# bench line of the same command: ms_per_step {B['ms_per_step']:.3f}, "
                    f"roofline.kernel_ms_avg {B['roofline']['kernel_ms_avg']:.3f} (HIP events), launches_per_step {B['roofline']['launches_per_step']}\n")
            f.write(f"{'kernel':72s} {'calls':>6s} {'total_ms':>12s} {'avg_ms':>12s} {'min_ms':>12s} {'max_ms':>12s} {'pct':>7s}\n")
            for r in ks:
                f.write(f"{r['Name'][:72]:72s} {int(r['Calls']):6d} {float(r['TotalDurationNs']) / 1e6:12.3f} {float(r['AverageNs']) / 1e6:12.4f} "
                        f"{float(r['MinNs']) / 1e6:12.4f} {float(r['MaxNs']) / 1e6:12.4f} {float(r['Percentage']):7.2f}\n")
            seen = set()
            f.write("\n# per-kernel resources (first dispatch)\n")
            for r in kt:
                if r["Kernel_Name"] in seen:
                    continue
                seen.add(r["Kernel_Name"])
                f.write(f"{r['Kernel_Name'][:72]:72s} grid={r['Grid_Size_X']} wg={r['Workgroup_Size_X']} lds={r['LDS_Block_Size']} "
                        f"scratch={r['Scratch_Size']} vgpr={r['VGPR_Count']} agpr={r['Accum_VGPR_Count']} sgpr={r['SGPR_Count']}\n")
            # timed launches only: the last steps x launches_per_step dispatches of the main kernel
            d = [(float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e6 for r in kt if kern in r["Kernel_Name"]]
            nl = int(round(B["steps"] * B["roofline"]["launches_per_step"]))
            if d and min(d) < 0.5 * max(d):
                small = [v for v in d if v < 0.5 * max(d)]
                f.write(f"\n# {kern}: the row above includes {len(small)} short launches ({min(small):.2f}-{max(small):.2f} ms): set_state's placement probes "
                        "(every chain pauses after 12 000 draws; csrc/pdmp_capi.hip init_state_tuned) -- not slices")
            if d:
                f.write(f"\n# {kern}: average over the {min(nl, len(d))} timed launches {sum(d[-nl:]) / len(d[-nl:]):.4f} ms "
                        f"(all {len(d)} launches incl. warm-up: {sum(d) / len(d):.4f} ms)\n")
        nl = int(round(B["steps"] * B["roofline"]["launches_per_step"]))
        pm = {}
        with open(os.path.join(summ, f"{tag}_{C}_pmc.txt"), "w") as f:
            f.write(f"# rocprofv3 --kernel-trace --pmc <counters> passes (one per line group) of the same bench command, config {C}\n"
                    "# means per dispatch; the headline rows use the timed launches only; WRITE_SIZE in KiB\n")
            for name in ("fetch", "write", "sq1", "sq2"):
                p = find(out, f"{C}_pmc", f"{name}_counter_collection.csv")
                allk = collections.defaultdict(lambda: collections.defaultdict(list))
                for r in rows(p):
                    allk[r["Kernel_Name"][:48]][r["Counter_Name"]].append(float(r["Counter_Value"]))
                for k, cs in sorted(allk.items()):
                    for cname, v in sorted(cs.items()):
                        timed = v[-nl:] if kern in k else v
                        f.write(f"{name:6s} {k:48s} {cname:22s} n={len(v)} mean_all={sum(v) / len(v):.6g} mean_timed={sum(timed) / len(timed):.6g}\n")
                        if kern in k:
                            pm[cname] = sum(timed) / len(timed)
            if "TCC_EA0_RDREQ_sum" in pm and "WRITE_SIZE" in pm:
                units_name = "event" if C == "C2" else "proposal"
                units = (B["totals"]["nevents"] if C == "C2" else B["proposals_per_s"] * B["ms_per_step"] * 1e-3 * B["steps"]) / max(nl, 1)
                rb, wb = pm["TCC_EA0_RDREQ_sum"] * 128.0, pm["WRITE_SIZE"] * 1024
                per = (rb + wb) / units
                alg = B["roofline"]["algorithmic_bytes_per_launch"]
                kms = B["roofline"]["kernel_ms_avg"]
                f.write(f"\n# HBM traffic per timed launch of {kern}: read {rb:.4g} B ({pm['TCC_EA0_RDREQ_sum']:.4g} line requests x 128 B) + written {wb:.4g} B "
                        f"(WRITE_SIZE x 1024) = {rb + wb:.4g} B = {per:.1f} B per {units_name} ({units:.4g} {units_name}s per launch, "
                        f"{pm['TCC_EA0_RDREQ_sum'] / units:.2f} lines read per {units_name}) = {(rb + wb) / (kms * 1e-3) / 1e12:.2f} TB/s at {kms:.2f} ms; "
                        f"algorithmic {alg:.4g} B per launch -> traffic / algorithmic = {(rb + wb) / alg:.2f}\n")
                if "TCC_HIT_sum" in pm:
                    f.write(f"# L2 hit rate {pm['TCC_HIT_sum'] / (pm['TCC_HIT_sum'] + pm['TCC_MISS_sum']):.3f}\n")
                traffic["configs"][C] = {"per": units_name, "hbm_bytes_per_unit": per, "read_bytes_per_launch": rb, "written_bytes_per_launch": wb,
                                         "source": f"profiles/{tag}_{C}_pmc.txt (accounting: profiles/{tag}_pmc_calibration.txt)"}
                if "SQ_INSTS_VALU" in pm and "SQ_WAVE_CYCLES" in pm:
                    # the instruction-issue account (SQ passes): per unit of work, and how a wavefront's cycles split
                    wc = pm["SQ_WAVE_CYCLES"]
                    iss = {"per": units_name, "valu": pm["SQ_INSTS_VALU"] / units, "salu": pm.get("SQ_INSTS_SALU", 0.0) / units,
                           "lds": pm.get("SQ_INSTS_LDS", 0.0) / units,
                           "vmem": (pm.get("SQ_INSTS_VMEM_RD", 0.0) + pm.get("SQ_INSTS_VMEM_WR", 0.0)) / units,
                           "waves": pm.get("SQ_WAVES"), "source": f"profiles/{tag}_{C}_pmc.txt (sq1, sq2 passes)"}
                    if "SQ_ACTIVE_INST_ANY" in pm:
                        iss.update(wave_cycles_issuing=pm["SQ_ACTIVE_INST_ANY"] / wc, wave_cycles_issuing_valu=pm.get("SQ_ACTIVE_INST_VALU", 0.0) / wc,
                                   wave_cycles_waiting_on_counters=pm.get("SQ_WAIT_ANY", 0.0) / wc,
                                   wave_cycles_waiting_for_issue=pm.get("SQ_WAIT_INST_ANY", 0.0) / wc)
                    if "SQ_BUSY_CYCLES" in pm and pm["SQ_BUSY_CYCLES"] > 0:
                        iss["waves_per_simd_avg"] = wc / pm["SQ_BUSY_CYCLES"] / 4.0  # (SQ_BUSY_CYCLES counts per SE-level SQ: see the pmc file for the raw values)
                    traffic["configs"][C]["issue"] = iss
                    f.write("# issue: " + json.dumps(iss) + "\n")
    json.dump(traffic, open(os.path.join(summ, "traffic.json"), "w"), indent=1)
    open(os.path.join(summ, f"{tag}_configs.jsonl"), "w").write("\n".join(lines) + "\n")
    print(open(os.path.join(summ, f"{tag}_pmc_calibration.txt")).read())
    print(json.dumps(traffic, indent=1))


if __name__ == "__main__":
    main()
