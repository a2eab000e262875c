#!/usr/bin/env python3
"""Turns the rocprofv3 runs of tools/profile_round.sh (gpurun_out/prof_<tag>/) into the tracked summaries
under profiles/:  <tag>_config<C>_kernel_stats.txt, <tag>_config<C>_pmc.txt and <tag>_counters_config<C>.json (what bench.py's
`roofline` object reads; it carries the hash of the kernel sources it was measured on).

HBM bytes per launch, since round 6 from the L2's memory-side requests BY SIZE (passes rdsz / wrsz of profile_round.sh):
read = 32 x TCC_EA0_RDREQ_32B + 64 x TCC_EA0_RDREQ_64B + 128 x TCC_EA0_RDREQ_128B (+ 64 x whatever TCC_EA0_RDREQ counts beyond
the three), written = 64 x TCC_EA0_WRREQ_64B + 32 x (TCC_EA0_WRREQ - TCC_EA0_WRREQ_64B).  Checked on a known stream: k_sort_a's
0.800 GB of events come out as 6.27 M requests of 128 bytes = 0.803 GB.  The figure of rounds 1-5 -- 2 x FETCH_SIZE + WRITE_SIZE
(KB = 1024 B; FETCH_SIZE tallies a 128-byte request at 64, MI355X_MICROARCH.md HBM section) -- is right for kernels that read
128-byte requests only and doubles everything else (a kernel of random 32- / 64-byte probes); it is kept beside the new one as
`hbm_bytes_fetch_x2_per_step`.  Either way requests that the Infinity Cache serves are counted (the counters sit on the L2's
side of it): a kernel whose figure exceeds the ~6.3 TB/s the HBM delivers says so (`above_hbm_achievable`).
Kernel -> phase: the library's roctx ranges (GX_ROCTX=1) in the trace pass; a kernel dispatch carries the stack id of the range
it was launched in.
"""
import glob
import json
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.environ.get("GX_PROFILE_OUT", os.path.join(ROOT, "profiles"))  # (on the GPU box: a directory under gpurun_out/, the only one that travels back)
sys.path.insert(0, ROOT)


def db_of(d):
    f = glob.glob(os.path.join(d, "**", "*.db"), recursive=True)
    return f[0] if f else None


def tables(c):
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    return lambda p: next(x for x in tabs if x.startswith(p))


def short(name):
    """k_xyz of a mangled gx:: kernel name (template instances share it); anything else as it is."""
    import re
    name = name.split("(")[0].replace("void ", "")
    m = re.match(r"_ZN2gx(\d+)", name)
    if m:
        n = int(m.group(1))
        return name[m.end():m.end() + n]
    m = re.match(r"gx::(k_[A-Za-z0-9_]+)", name)
    if m:
        return m.group(1)
    return name[:60]


def kernel_times(db):
    c = sqlite3.connect(db)
    t = tables(c)
    kd, ks = t("rocpd_kernel_dispatch"), t("rocpd_info_kernel_symbol")
    agg = {}
    for name, a, b in c.execute(f"select s.kernel_name, d.start, d.end from {kd} d join {ks} s on d.kernel_id = s.id"):
        agg.setdefault(name.split("(")[0].replace("void ", ""), []).append(b - a)
    return agg


def kernel_phases(db):
    """kernel (short name) -> the library phase it is launched in ("tile" of "gx:t.tile"), from the marker regions of the trace."""
    c = sqlite3.connect(db)
    t = tables(c)
    try:
        kd, ks, ev, rg = t("rocpd_kernel_dispatch"), t("rocpd_info_kernel_symbol"), t("rocpd_event"), t("rocpd_region")
    except StopIteration:
        return {}
    stack = {}
    for sid, ext in c.execute(f"select e.stack_id, e.extdata from {rg} r join {ev} e on r.event_id = e.id"):
        msg = json.loads(ext or "{}").get("message", "")
        if msg.startswith("gx:"):
            name = msg[3:]
            stack[sid] = name[2:] if len(name) > 2 and name[1] == "." else name
    votes = {}
    for name, sid in c.execute(f"select s.kernel_name, e.stack_id from {kd} d join {ks} s on d.kernel_id = s.id join {ev} e on d.event_id = e.id"):
        if sid in stack:
            v = votes.setdefault(short(name), {})
            v[stack[sid]] = v.get(stack[sid], 0) + 1
    return {k: max(v, key=v.get) for k, v in votes.items()}


def pmc_means(db):
    c = sqlite3.connect(db)
    t = tables(c)
    pe, kd, ks, ip = t("rocpd_pmc_event"), t("rocpd_kernel_dispatch"), t("rocpd_info_kernel_symbol"), t("rocpd_info_pmc")
    agg = {}
    for name, ctr, v in c.execute(
            f"select s.kernel_name, i.name, e.value from {pe} e join {kd} d on e.event_id = d.event_id "
            f"join {ks} s on d.kernel_id = s.id join {ip} i on e.pmc_id = i.id"):
        agg.setdefault((name.split("(")[0].replace("void ", ""), ctr), []).append(v)
    return agg


def main():
    tag = sys.argv[1]
    cfg = sys.argv[2] if len(sys.argv) > 2 else "2"   # (2, 3, 4, 5 or a name: 2E)
    src = sys.argv[3] if len(sys.argv) > 3 else os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
    from bench import source_hash
    steps = 5  # profile_round.sh: --warmup 1 --steps 2, and bench.py's two extra steps that time every phase
    out = {"_how": __doc__.strip().split("\n\n")[1].replace("\n", " "), "tag": tag, "config": int(cfg) if cfg.isdigit() else cfg,
           "source_hash": source_hash(),
           "kernels": {}, "whole_step": {}}
    # kernel durations
    db = db_of(os.path.join(src, "trace"))
    lines = []
    if db:
        agg = kernel_times(db)
        tot = sum(sum(v) for v in agg.values())
        lines.append(f"{'kernel':60s} {'calls':>6s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'pct':>6s}")
        for name, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
            lines.append(f"{name[:60]:60s} {len(v):6d} {sum(v)/1e6:10.3f} {sum(v)/len(v)/1e3:10.2f} {min(v)/1e3:10.2f} "
                         f"{max(v)/1e3:10.2f} {100*sum(v)/tot:6.2f}")
            k = out["kernels"].setdefault(short(name), {})
            k["us_per_step"] = k.get("us_per_step", 0.0) + sum(v) / 1e3 / steps
            k["launches_per_step"] = k.get("launches_per_step", 0.0) + len(v) / steps
        out["whole_step"]["kernel_ms_per_step"] = tot / 1e6 / steps
        ph = kernel_phases(db)
        for k, p in ph.items():
            out["kernels"].setdefault(k, {})["phase"] = p
        by_phase = {}
        for k, v in out["kernels"].items():
            by_phase.setdefault(v.get("phase", "(none)"), []).append((v.get("us_per_step", 0.0), k))
        lines.append("")
        lines.append("phase (roctx range of the library) : kernels launched in it, us per step")
        for p, ks_ in sorted(by_phase.items(), key=lambda kv: -sum(x[0] for x in kv[1])):
            lines.append(f"  {p:10s} {sum(x[0] for x in ks_):9.1f}  " + ", ".join(f"{k} {u:.1f}" for u, k in sorted(ks_, reverse=True) if u >= 0.05))
        open(os.path.join(OUT, f"{tag}_config{cfg}_kernel_stats.txt"), "w").write("\n".join(lines) + "\n")
    # counters
    pm = {}
    for p in ("fetch", "write", "rdsz", "wrsz", "sq1", "sq2"):
        db = db_of(os.path.join(src, p))
        if db:
            pm.update(pmc_means(db))
    lines = [f"{'kernel':58s} {'counter':22s} {'calls':>6s} {'mean':>18s}"]
    for (name, ctr), v in sorted(pm.items(), key=lambda kv: (kv[0][1], -sum(kv[1]) / len(kv[1]))):
        lines.append(f"{name[:58]:58s} {ctr:22s} {len(v):6d} {sum(v)/len(v):18.1f}")
    open(os.path.join(OUT, f"{tag}_config{cfg}_pmc.txt"), "w").write("\n".join(lines) + "\n")
    fetch_tot = write_tot = 0.0
    per = {}
    for (name, ctr), v in pm.items():
        per.setdefault(name, {})[ctr] = (sum(v) / len(v), len(v))
    for name, cs in per.items():
        k = out["kernels"].setdefault(short(name), {})
        f = cs.get("FETCH_SIZE", (0, 0))
        w = cs.get("WRITE_SIZE", (0, 0))
        fetch_tot += f[0] * f[1]
        write_tot += w[0] * w[1]
        if "FETCH_SIZE" in cs or "WRITE_SIZE" in cs:
            k["fetch_kb_raw_per_step"] = k.get("fetch_kb_raw_per_step", 0) + f[0] * f[1] / steps
            k["write_kb_raw_per_step"] = k.get("write_kb_raw_per_step", 0) + w[0] * w[1] / steps
            k["hbm_bytes_fetch_x2_per_step"] = k.get("hbm_bytes_fetch_x2_per_step", 0) + (2 * f[0] * f[1] + w[0] * w[1]) * 1024 / steps
        tot_of = lambda c: cs[c][0] * cs[c][1] / steps if c in cs else None  # noqa: E731
        if tot_of("TCC_EA0_RDREQ_sum") is not None:
            r, r32, r64, r128 = (tot_of(c) or 0.0 for c in ("TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_EA0_RDREQ_64B_sum", "TCC_EA0_RDREQ_128B_sum"))
            k["read_requests_per_step"] = {"all": k.get("read_requests_per_step", {}).get("all", 0) + r,
                                           "32B": k.get("read_requests_per_step", {}).get("32B", 0) + r32,
                                           "64B": k.get("read_requests_per_step", {}).get("64B", 0) + r64,
                                           "128B": k.get("read_requests_per_step", {}).get("128B", 0) + r128}
            k["read_bytes_per_step"] = k.get("read_bytes_per_step", 0) + 32 * r32 + 64 * r64 + 128 * r128 + 64 * max(0.0, r - r32 - r64 - r128)
        if tot_of("TCC_EA0_WRREQ_sum") is not None:
            wq, w64 = tot_of("TCC_EA0_WRREQ_sum") or 0.0, tot_of("TCC_EA0_WRREQ_64B_sum") or 0.0
            k["written_bytes_per_step"] = k.get("written_bytes_per_step", 0) + 64 * w64 + 32 * max(0.0, wq - w64)
        # SQ counters: totals per step over the template instances of one kernel
        for c in cs:
            if c.startswith("SQ_"):
                k.setdefault("sq", {})
                k["sq"][c] = k["sq"].get(c, 0.0) + cs[c][0] * cs[c][1] / steps
    # a kernel's HBM bytes: by request size when those passes ran, the FETCH_SIZE x 2 figure otherwise
    by_size = any("read_bytes_per_step" in v for v in out["kernels"].values())
    for kn, v in out["kernels"].items():
        if "read_bytes_per_step" in v or "written_bytes_per_step" in v:
            v["hbm_bytes_per_step"] = v.get("read_bytes_per_step", 0.0) + v.get("written_bytes_per_step", 0.0)
        elif "hbm_bytes_fetch_x2_per_step" in v:
            v["hbm_bytes_per_step"] = v["hbm_bytes_fetch_x2_per_step"]
        if v.get("hbm_bytes_per_step") and v.get("us_per_step"):
            v["tbs"] = v["hbm_bytes_per_step"] / v["us_per_step"] / 1e6
            if v["tbs"] > 6.3:
                v["above_hbm_achievable"] = "more than the ~6.3 TB/s HBM delivers: part of these requests were served by the Infinity Cache"
    out["whole_step"]["fetch_kb_raw"] = fetch_tot / steps
    out["whole_step"]["write_kb_raw"] = write_tot / steps
    out["whole_step"]["hbm_bytes_fetch_x2_per_step"] = (2 * fetch_tot + write_tot) * 1024 / steps
    out["whole_step"]["hbm_bytes_per_step"] = (sum(v.get("hbm_bytes_per_step", 0.0) for v in out["kernels"].values()) if by_size
                                               else (2 * fetch_tot + write_tot) * 1024 / steps)
    out["whole_step"]["bytes_from"] = "requests by size (TCC_EA0_RDREQ_32B/64B/128B, TCC_EA0_WRREQ[_64B])" if by_size else "2 x FETCH_SIZE + WRITE_SIZE"
    # the dominant kernel = the one with the most time per step; its issue model from the SQ counters (quad-cycle
    # units, MI355X_MICROARCH.md)
    timed = {k: v for k, v in out["kernels"].items() if v.get("us_per_step") and k.startswith("k_")}
    domk = max(timed, key=lambda k: timed[k]["us_per_step"]) if timed else None
    if domk:
        d = timed[domk]
        out["dominant"] = {"kernel": domk, "us_per_step": d["us_per_step"], "launches_per_step": d.get("launches_per_step"),
                           "hbm_bytes_per_step": d.get("hbm_bytes_per_step"),
                           "tbs": (d["hbm_bytes_per_step"] / d["us_per_step"] / 1e6) if d.get("hbm_bytes_per_step") else None}
    kt = out["kernels"].get(domk, {}) if domk else {}
    sq = kt.get("sq", {})
    if sq.get("SQ_WAVE_CYCLES") and sq.get("SQ_BUSY_CYCLES"):
        wc = sq["SQ_WAVE_CYCLES"]
        out["issue"] = {
            "kernel": domk,
            "valu_insts_per_wave": sq.get("SQ_INSTS_VALU", 0) / max(1.0, sq.get("SQ_WAVES", 1)),
            "lds_insts_per_wave": sq.get("SQ_INSTS_LDS", 0) / max(1.0, sq.get("SQ_WAVES", 1)),
            "active_valu_frac_of_wave_cycles": sq.get("SQ_ACTIVE_INST_VALU", 0) / wc,
            "active_lds_frac_of_wave_cycles": sq.get("SQ_ACTIVE_INST_LDS", 0) / wc,
            "wait_any_frac_of_wave_cycles": sq.get("SQ_WAIT_ANY", 0) / wc if "SQ_WAIT_ANY" in sq else None,
            "wait_inst_any_frac_of_wave_cycles": sq.get("SQ_WAIT_INST_ANY", 0) / wc if "SQ_WAIT_INST_ANY" in sq else None,
            "wait_inst_lds_frac_of_wave_cycles": sq.get("SQ_WAIT_INST_LDS", 0) / wc,
            "lds_bank_conflict_frac_of_lds_cycles": (sq.get("SQ_LDS_BANK_CONFLICT", 0) / sq["SQ_LDS_IDX_ACTIVE"])
            if sq.get("SQ_LDS_IDX_ACTIVE") else None,
            "raw": sq,
        }
    json.dump(out, open(os.path.join(OUT, f"{tag}_counters_config{cfg}.json"), "w"), indent=1)
    print(json.dumps({k: v for k, v in out.items() if k != "kernels"}, indent=1)[:3000])


if __name__ == "__main__":
    main()
