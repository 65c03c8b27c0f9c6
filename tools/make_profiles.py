"""Copy the judged evidence of tools/final_prof.sh (gpurun_out/final/) into profiles/ (tracked):
rocprofv3 kernel stats, bench lines, PMC summaries and the per-launch HBM traffic table bench.py reads."""
import json, os, re, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "final")
DST = os.path.join(ROOT, "profiles")
if len(sys.argv) < 2:
    raise SystemExit("usage: make_profiles.py rNN (the tag given to tools/final_prof.sh)")
tag = sys.argv[1]

def pmc(path, counter):
    out, name = {}, None
    for line in open(path):
        if not line.startswith(" "):
            name = line.strip()
        elif counter in line and name:
            out[name] = float(line.split("mean=")[1])
    return out

fetch = pmc(os.path.join(SRC, "pmc_fetch.txt"), "FETCH_SIZE")
write = pmc(os.path.join(SRC, "pmc_write.txt"), "WRITE_SIZE")
# (pattern, bench.py kernel name): first match wins, the more specific pattern first
short = [("block_fwd_rs_kernel", "block_fwd"), ("block_bwd_attn", "block_bwd_attn"),
         ("block_bwd_ln1mlp", "block_bwd_ln1mlp"), ("block_bwd_ln1", "block_bwd_ln1"),
         ("block_bwd_mlp", "block_bwd_mlp"), ("tokenize_bwd", "tokenize_bwd"), ("tokenize_fwd", "tokenize_fwd"),
         ("head_bwd", "head_bwd"), ("reduce_segs", "reduce_slabs"), ("adamw_kernel", "adamw"), ("head_fwd", "head_fwd"),
         ("prep_weights", "prep_weights")]
kern = {}
for full, f in fetch.items():
    for pat, s in short:
        if pat in full:
            w = write.get(full, 0.0)
            if s not in kern:
                kern[s] = {"FETCH_SIZE_KB": f, "WRITE_SIZE_KB": w, "hbm_bytes_per_launch": int((2 * f + w) * 1024)}
            break
json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), bench.py --steps 1 --warmup 1 "
                     "--batch 256 (tools/final_prof.sh), final kernels of the round",
           "correction": "FETCH_SIZE x2 (gfx950 16B/lane streaming under-report, MI355X_MICROARCH.md), WRITE_SIZE x1; KB -> bytes",
           "kernels": kern}, open(os.path.join(DST, f"{tag}_pmc_traffic.json"), "w"), indent=1)
if "--traffic-only" in sys.argv:
    raise SystemExit(0)
shutil.copy(os.path.join(SRC, "stats", f"{tag}_kernel_stats.csv"), os.path.join(DST, f"{tag}_kernel_stats_bench_b256.csv"))
shutil.copy(os.path.join(SRC, "bench_default.json"), os.path.join(DST, f"{tag}_bench_default.json"))
shutil.copy(os.path.join(SRC, "bench_profile_all.json"), os.path.join(DST, f"{tag}_bench_b256_events.json"))
shutil.copy(os.path.join(SRC, "bench_other_configs.jsonl"), os.path.join(DST, f"{tag}_bench_other_configs.jsonl"))
with open(os.path.join(DST, f"{tag}_pmc_summary.txt"), "w") as f:
    f.write("# rocprofv3 --pmc passes (SQ / LDS counters, then HBM FETCH_SIZE / WRITE_SIZE in KB) of bench.py --steps 1 --warmup 1\n")
    for n in ("pmc_sq.txt", "pmc_fetch.txt", "pmc_write.txt"):
        f.write(open(os.path.join(SRC, n)).read())
for n, dst in (("dp_overlap.txt", f"{tag}_dp_overlap.txt"), ("parity_measured.jsonl", f"{tag}_parity_measured.jsonl"),
               ("gpu_tests.txt", f"{tag}_gpu_tests.txt"), ("peak_microbench.json", f"{tag}_peak_microbench.json"),
               ("cu_contention.jsonl", f"{tag}_dp_cu_contention.jsonl"), ("gate_microbench.jsonl", f"{tag}_gate_microbench.jsonl"),
               ("fwd_stack_ab.txt", f"{tag}_fwd_stack_ab.txt"), ("gate_qkv.txt", f"{tag}_gate_qkv.txt")):
    # (gpurun MERGES the box's files into gpurun_out/: a file this pass did not write may be a leftover of an earlier round -- skip it)
    fresh = os.path.getmtime(os.path.join(SRC, "bench_default.json")) - 3600
    if os.path.exists(os.path.join(SRC, n)) and os.path.getmtime(os.path.join(SRC, n)) >= fresh:
        shutil.copy(os.path.join(SRC, n), os.path.join(DST, dst))
print(json.dumps(kern, indent=1))
