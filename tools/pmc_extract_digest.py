"""profiles/rNN_extractor_counters.json (round tag: argv[3], default r05) from the PMC summary (tools/pmc_summary2.py output) and the kernel stats of
tools/prof_extract.py 512: python tools/pmc_extract_digest.py <pmc txt> <kernel_stats.csv>"""
import csv, json, re, sys
txt = open(sys.argv[1]).read()
stats = {r["Name"].split("(")[0].split("::")[-1].split("<")[0]: float(r["AverageNs"]) / 1e3 for r in csv.DictReader(open(sys.argv[2]))}   # (templates: name<..>)
tag = sys.argv[3] if len(sys.argv) > 3 else "r06"
batch = int(sys.argv[4]) if len(sys.argv) > 4 else 512
out = {"source": "profiles/%s_pmc_extract_b512.txt (rocprofv3 --pmc passes of tools/prof_extract.py 512, tools/prof_%s.sh) + "
                 "profiles/r02_hbm_counter_calibration.txt" % (tag, tag), "batch": batch}
for k in ("fast_cells_kernel", "describe_kernel"):
    d = {}
    for line in txt.splitlines():
        if k in line:
            for name, val in re.findall(r"(\w+) avg ([0-9.e+\-]+)", line):
                d[{"FETCH_SIZE": "FETCH_SIZE_KB", "WRITE_SIZE": "WRITE_SIZE_KB"}.get(name, name)] = float(val)
    d["kernel_us"] = stats.get(k, 0.0)
    if not d["kernel_us"]:
        raise SystemExit("pmc_extract_digest: no kernel named %s in %s (names: %s)" % (k, sys.argv[2], sorted(stats)))
    out[k] = {x: d[x] for x in ("FETCH_SIZE_KB", "WRITE_SIZE_KB", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "GRBM_GUI_ACTIVE", "SQ_WAVES", "kernel_us") if x in d}
out["calibration"] = {"FETCH_SIZE_factor_aligned": 2.0, "FETCH_SIZE_factor_unaligned_32bit": 16.0 / 9.0, "WRITE_SIZE_factor": 1.0,
                      "note": "1 GiB copies (4 x the Infinity Cache): FETCH_SIZE reports 0.500 of the bytes of aligned reads of any width, "
                              "0.5625 of 4-byte reads at byte offset 1; WRITE_SIZE reports 1.000"}
print(json.dumps(out, indent=1))
