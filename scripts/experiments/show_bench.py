"""Print the headline fields of a bench.py JSON line (file argument)."""
import json, sys
d = json.loads([ln for ln in open(sys.argv[1]) if ln.startswith("{")][-1])
print({k: d[k] for k in ("value", "ms_per_step", "scaling", "steps", "warmup")})
print(d["roofline"])
print(d.get("cpu_baseline"))
print(d.get("cpu_baseline_blas_search_only"))
print(d.get("extra_error"), d.get("cpu_baseline_error"))
e = d.get("extra", {}).get("config2_shape", {})
print({k: e.get(k) for k in ("ms_per_step", "query_videos_per_s", "int8_prefilter_tops", "score_normalize_queries_ms",
                             "value_with_score_norm", "set_queries_ms", "kernel_ms_per_step")})
print(e.get("knn_200k_x_2M"))
print(e.get("roofline_fp32_route"))
for k, v in d["kernels"].items():
    print(k[:60], {a: (round(b, 3) if isinstance(b, float) else b) for a, b in v.items() if a != "note"})
