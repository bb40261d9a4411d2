"""Copy the evidence set tools/gpu_evidence.sh left under gpurun_out/<tag>/ into profiles/ (tracked), named per round, and rebuild the
three pmc_traffic_*.json files bench.py attaches as roofline.traffic.
    python tools/install_evidence.py r03 r03          # <tag under gpurun_out> <prefix in profiles/>"""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def lib_sha256():
    import hashlib
    return hashlib.sha256(open(os.path.join(ROOT, "a-loam_amd", "lib", "libaloam_mi355x.so"), "rb").read()).hexdigest()


def main(tag, prefix):
    src, dst = os.path.join(ROOT, "gpurun_out", tag), os.path.join(ROOT, "profiles")
    # one binary for the whole set: the counters, the kernel stats and the bench line must be of the library that is in the tree now
    here = lib_sha256()
    line0 = json.loads(open(os.path.join(src, "bench.log")).read().strip().splitlines()[-1])
    tr0 = json.load(open(os.path.join(src, "pmc_headline", "pmc_traffic.json")))
    for what, sha in (("bench line", line0.get("library_sha256")), ("counter passes", tr0.get("lib_sha256"))):
        if sha != here:
            raise SystemExit(f"refusing to install: the {what} under gpurun_out/{tag} are of library {str(sha)[:12]}, a-loam_amd/lib holds {here[:12]}")
    cp = lambda a, b: shutil.copyfile(os.path.join(src, a), os.path.join(dst, b))
    line = open(os.path.join(src, "bench.log")).read().strip().splitlines()[-1]
    batch = json.loads(line)["config"]["sequences_per_gpu"]
    with open(os.path.join(dst, f"{prefix}_bench_b{batch}.json"), "w") as f:
        json.dump(json.loads(line), f, indent=1)
    cp("pytest_gpu.log", f"{prefix}_pytest_gpu.log")
    for name in ("two_rank_shared_gpu.json", "eight_rank_shared_gpu.json", "launcher_rccl_one_rank.json"):   # written by the -m gpu tests of the same call
        if os.path.exists(os.path.join(ROOT, "gpurun_out", name)):
            shutil.copyfile(os.path.join(ROOT, "gpurun_out", name), os.path.join(dst, f"{prefix}_{name}"))
    map_batch = next((w["sequences_per_gpu"] for k, w in json.loads(line).get("workloads", {}).items() if "steady-state" in k), 256)   # --mapping = the travelling workload
    side_batch = next((w["sequences_per_gpu"] for k, w in json.loads(line).get("workloads", {}).items() if k.startswith("configs[3]")), 1024)   # --sensor ROWS128 / --travel runs of gpu_evidence.sh: --batch 1024
    batch_of = {"headline": batch, "mapping": map_batch, "rows128": side_batch, "travel": side_batch}
    for cfg in ("headline", "mapping", "rows128", "travel"):
        if os.path.exists(os.path.join(src, f"kernel_stats_{cfg}.md")):
            cp(f"kernel_stats_{cfg}.md", f"{prefix}_kernel_stats_{cfg}_b{batch_of[cfg]}.md")
    for extra in ("pmc_mapping_sq1.md", "pmc_mapping_tcp.md"):
        if os.path.exists(os.path.join(src, extra)):
            cp(extra, f"{prefix}_{extra[:-3]}_b{map_batch}.md")
    for c in ("fetch", "write", "sq1", "sq2"):
        cp(f"pmc_headline/pmc_{c}.md", f"{prefix}_pmc_{c}_headline_b{batch}.md")
    traffic = {"headline": json.load(open(os.path.join(src, "pmc_headline", "pmc_traffic.json")))}
    traffic["headline"]["source"] += f"; profiles/{prefix}_pmc_fetch_headline_b{batch}.md, {prefix}_pmc_write_headline_b{batch}.md"
    for cfg, args, sensor in (("mapping", "--mapping", "HDL-64"), ("rows128", "--sensor ROWS128", "ROWS128")):
        for c in ("FETCH_SIZE", "WRITE_SIZE"):
            cp(f"pmc_{cfg}_{c}.md", f"{prefix}_pmc_{c.split('_')[0].lower()}_{cfg}_b{batch_of[cfg]}.md")
        fj, wj = (json.load(open(os.path.join(src, f"pmc_{cfg}_{c}.json"))) for c in ("FETCH_SIZE", "WRITE_SIZE"))
        traffic[cfg] = {"batch": batch_of[cfg], "mapping": cfg == "mapping", "sensor": sensor, "lib_sha256": here,
                        "source": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes of python bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 {args} "
                                  f"(tools/gpu_evidence.sh); profiles/{prefix}_pmc_fetch_{cfg}_b{batch_of[cfg]}.md, {prefix}_pmc_write_{cfg}_b{batch_of[cfg]}.md",
                        "fetch_kib": {k: v["FETCH_SIZE"] for k, v in fj.items() if "FETCH_SIZE" in v}, "write_kib": {k: v["WRITE_SIZE"] for k, v in wj.items() if "WRITE_SIZE" in v},
                        "avg_us": {k: v.get("avg_us") for k, v in fj.items()}}
    for cfg, name in (("headline", "pmc_traffic_latest.json"), ("mapping", "pmc_traffic_mapping.json"), ("rows128", "pmc_traffic_rows128.json")):
        json.dump(traffic[cfg], open(os.path.join(dst, name), "w"), indent=1)
    # per-step traffic table: 2 x FETCH + WRITE per dispatch x dispatches per step
    for cfg in ("headline", "mapping", "rows128"):
        t = traffic[cfg]
        rows = sorted(((2 * t["fetch_kib"][k] + t["write_kib"].get(k, 0.0)) * 1024, k) for k in t["fetch_kib"])
        print(cfg, "per-dispatch traffic (GB), largest first:", [(k, round(b / 1e9, 3)) for b, k in rows[::-1][:10]])


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
