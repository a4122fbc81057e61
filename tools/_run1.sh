cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3e
python bench.py --steps 20 --warmup 3 > gpurun_out/r3e/bench.json 2> gpurun_out/r3e/bench.err; echo "rc bench $?"
python - <<'PY'
import json
r = json.load(open("gpurun_out/r3e/bench.json"))
print({k: r.get(k) for k in ("value", "ms_per_step", "warm_mvms_per_s", "cold_mvms_per_s", "build_ms", "config3_cg_ms", "config5_mvm_us")})
print({k: (v["us_per_mvm"], v["frac"], v["traffic_MB_per_launch"]) for k, v in r["stages"].items()})
print(r["roofline"])
print(r["lattice_row_order"]["warm_mvms_per_s"], {k: (v["us_per_mvm"], v["frac"]) for k, v in r["lattice_row_order"]["stages"].items()})
print(r["fine"]["warm_mvms_per_s"], r["fine"]["blur_roofline"], r["fine"]["build_ms"])
print(r.get("config3"), r.get("config4"), r.get("config5"))
PY
