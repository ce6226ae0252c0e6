"""profiles/r06_lba_counters.json from the PMC summary of the mixed 64-window LocalBA batch (tools/pmc_summary2.py output: FETCH_SIZE /
WRITE_SIZE in KB per launch, the MFMA counters) and the kernel stats of the same command:
    python tools/pmc_lba_digest.py <pmc_lba.txt> <lba_het64_kernel_stats.csv>"""
import csv, json, re, sys
txt = open(sys.argv[1]).read()
stats = {r["Name"].split("(")[0].split("::")[-1].replace("void ", ""): (float(r["AverageNs"]) / 1e3, int(r["Calls"])) for r in csv.DictReader(open(sys.argv[2]))}
out = {"source": "profiles/r06_pmc_lba.txt + profiles/r06_lba_het64_kernel_stats.csv (rocprofv3 passes of LBA_MIX=het LBA_N=64 tools/gpu_lba_mix_prof.py, "
                 "tools/prof_r06.sh); FETCH_SIZE / WRITE_SIZE in KB per launch, corrected like the extractor's (profiles/r02_hbm_counter_calibration.txt: "
                 "FETCH_SIZE reports half the bytes of aligned reads)", "windows": 64, "kernels": {}}
for k in ("k_schur", "k_points_walk", "k_lin<true>", "k_ldlt_reg", "k_ldlt_dev", "k_lm_init", "k_transition", "k_prepare", "k_final"):
    d = {}
    for line in txt.splitlines():
        if ("::" + k + " ") in line + " " or line.strip().startswith(("aos2::" + k, "void aos2::" + k)):
            for name, val in re.findall(r"(\w+) avg ([0-9.e+\-]+)", line):
                d[name] = float(val)
    if not d:
        continue
    us, calls = stats.get(k, (0.0, 0))
    d["kernel_us"], d["launches"] = us, calls
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d and us > 0:
        d["hbm_bytes_per_launch"] = (2.0 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024.0
        d["hbm_GB_per_s"] = d["hbm_bytes_per_launch"] / (us * 1e-6) / 1e9
    if "SQ_VALU_MFMA_BUSY_CYCLES" in d and d.get("SQ_BUSY_CYCLES"):
        d["mfma_busy_over_sq_busy"] = d["SQ_VALU_MFMA_BUSY_CYCLES"] / d["SQ_BUSY_CYCLES"]
    out["kernels"][k] = d
print(json.dumps(out, indent=1))
