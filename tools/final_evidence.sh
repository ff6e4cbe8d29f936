cd $GRAFT_REPO_ROOT
timeout -k 10 500 bash tools/profile_round.sh r1 > gpurun_out/pr.log 2>&1; tail -1 gpurun_out/pr.log | cut -c1-80
timeout -k 10 300 bash tools/profile_belief.sh r1 20000 > gpurun_out/pb.log 2>&1; tail -1 gpurun_out/pb.log | cut -c1-80
timeout -k 10 500 python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err
python3 - <<'PY'
import json
d=json.loads(open("gpurun_out/bench_final.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["config"]["single_query"])
print({k:v for k,v in d["roofline"].items() if k in ("kernel","achieved","frac","traffic","avg_launch_us")})
print(d["cpu_baseline"]["value"])
b=d["config"]["belief_space"]
print(b["build_belief_graph"]["ms_wall"], b["expected_costs"]["ms_wall"], b["expected_costs"]["sweeps"], b["extract_policy"]["ms_wall"], d["config"]["prm_roadmap"]["ms_wall"])
PY
