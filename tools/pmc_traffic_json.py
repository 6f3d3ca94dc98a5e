"""profiles/rNN_pmc_traffic_nn_compact.json from a tools/profile_bench.sh PMC summary (pmc_summary.py --json).

Usage: python tools/pmc_traffic_json.py gpurun_out/<tag>_pmc_summary.json gpurun_out/<tag>_under_rocprof.json [round] > profiles/r04_pmc_traffic_nn_compact.json
The second file is the bench line of the same command (algorithmic bytes and jobs per launch come from it).
"""
import json, sys

pmc = json.load(open(sys.argv[1]))
bench = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
# the warm-pass instantiation (moments, not pairs): the one with the most launches
k = max((n for n in pmc if n.startswith("void gloc::reg::nn_compact_kernel<2, false")), key=lambda n: pmc[n]["dispatches"])
c = pmc[k]
roof = bench["roofline"]
fetch_kb, write_kb = c["FETCH_SIZE"], c["WRITE_SIZE"]
hbm = (2.0 * fetch_kb + write_kb) * 1024.0
waves = c["SQ_WAVES"]
# SQ_BUSY_CYCLES sums the 32 shader engines; SQ_ACTIVE_INST_VALU counts quad-cycles over the 1024 SIMDs
busy_cycles = c["SQ_BUSY_CYCLES"] / 32.0
out = {
    "round": int(sys.argv[3]) if len(sys.argv) > 3 else 4,
    "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE / SQ_* / TCC_HIT_sum TCC_MISS_sum (separate passes, "
              "tools/profile_bench.sh) over `python3 bench.py --no-cpu-baseline --no-legs --steps 5 --warmup 2 --reps 1`: "
              f"{c['dispatches']} launches of {roof['jobs_per_launch']:.0f} jobs each; written by tools/pmc_traffic_json.py",
    "kernel": k.replace("void ", ""),
    "jobs_per_launch": roof["jobs_per_launch"],
    "fetch_size_kb_per_launch": fetch_kb,
    "write_size_kb_per_launch": write_kb,
    "correction": "gfx950: FETCH_SIZE reads half the bytes of 16-B-per-lane streams (MI355X_MICROARCH.md, HBM section) -> doubled "
                  "(every read of this kernel is a 16-B-per-lane load); WRITE_SIZE taken as is; L2 -> fabric requests, "
                  "Infinity-Cache hits included",
    "hbm_bytes_per_launch": hbm,
    "algorithmic_bytes_per_launch": roof["algorithmic_bytes_per_launch"],
    "traffic_over_algorithmic": hbm / roof["algorithmic_bytes_per_launch"],
    "l2_hit_rate": c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]),
    "mean_us_under_pmc": c["mean_us_under_pmc"],
    "per_wave": {"waves": waves, "valu": c["SQ_INSTS_VALU"] / waves, "salu": c["SQ_INSTS_SALU"] / waves,
                 "lds": c["SQ_INSTS_LDS"] / waves, "vmem_rd": c["SQ_INSTS_VMEM_RD"] / waves,
                 "wave_cycles": 4.0 * c["SQ_WAVE_CYCLES"] / waves},
    "valu_busy_frac": 4.0 * c["SQ_ACTIVE_INST_VALU"] / 1024.0 / busy_cycles,
    "valu_busy_note": "SQ_ACTIVE_INST_VALU (quad-cycles, summed over 1024 SIMDs) x 4 / 1024 over SQ_BUSY_CYCLES / 32 shader engines",
    "job_group": 24,
    "database_scans_target_index": bench["config"].get("database_scans_target_index"),
}
print(json.dumps(out, indent=1))
