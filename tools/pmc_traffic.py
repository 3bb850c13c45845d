"""Aggregate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into profiles/hbm_traffic.json.

    python tools/pmc_traffic.py <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass> <out.json> [note]

Each pass is `rocprofv3 --pmc X --kernel-trace --output-format csv -d <dir> -- python3 bench.py
--steps 1 --warmup 0 --no-cpu`; the counter values are in KiB (x 1024 = bytes), summed over all
launches of a kernel in the pass and divided by the launch count."""
import csv, glob, json, re, sys


def base_name(n):
    n = n.replace("(anonymous namespace)::", "")
    n = re.sub(r"^void\s+", "", n)
    m = re.match(r"([A-Za-z_0-9]+)(<[^(]*>)?\(", n)
    if not m:
        return n.split("(")[0]
    name, targs = m.group(1), m.group(2) or ""
    if name == "k_chain_finish":
        return "k_chain_finish<lds256>" if targs.startswith("<256") else "k_chain_finish<global>"
    return name


def collect(d, counter):
    tot, cnt = {}, {}
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            k = base_name(r["Kernel_Name"])
            tot[k] = tot.get(k, 0.0) + float(r["Counter_Value"]) * 1024.0
            cnt[k] = cnt.get(k, 0) + 1
    return tot, cnt


def main():
    fdir, wdir, out = sys.argv[1:4]
    note = sys.argv[4] if len(sys.argv) > 4 else ""
    ft, fc = collect(fdir, "FETCH_SIZE")
    wt, wc = collect(wdir, "WRITE_SIZE")
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from flye_amd import config
    # bench.py only quotes these figures while the kernel sources and the detector configuration are the same
    res = {"_note": note, "_stamp": {"source_sha256": bench.kernel_source_digest(),
                                     "min_overlap": config.DETECTOR_MIN_OVERLAP}}
    for k in sorted(set(ft) | set(wt)):
        n = max(fc.get(k, 0), wc.get(k, 0))
        tot = ft.get(k, 0.0) + wt.get(k, 0.0)
        res[k] = {"FETCH_SIZE_bytes": int(ft.get(k, 0)), "WRITE_SIZE_bytes": int(wt.get(k, 0)),
                  "bytes_per_pass": int(tot), "launches_profiled": n, "bytes_per_launch": int(tot / max(1, n))}
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)
    for k, v in sorted(res.items(), key=lambda kv: -(kv[1].get("bytes_per_pass", 0) if isinstance(kv[1], dict) else 0))[:12]:
        if isinstance(v, dict) and "bytes_per_pass" in v:
            print(f"{k:28s} {v['bytes_per_pass'] / 1e9:8.3f} GB/pass  {v['launches_profiled']:3d} launches")


if __name__ == "__main__":
    main()
