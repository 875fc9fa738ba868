#!/bin/bash
# via gpurun: whole GPU suite, then the beam-100 legs with the single-block selection (default below 256 beams) and with the
# radix selection forced. Usage: tools/r06_round4.sh TAG
TAG=${1:-r06e}
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 2700 python -m pytest tests -m gpu -q --maxfail=10 -s > $O/pytest.log 2>&1
echo "pytest rc=$?" | tee -a $O/pytest.log
grep -E "passed|failed|error|^FAILED|^ERROR" $O/pytest.log | tail -30
for r in default 1; do
  if [ $r = 1 ]; then export RPR_SELECT_RADIX=1; fi
  timeout 1500 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-exact-fp32 --no-roofline --secondary config4,rankdata_ref_flags,f2 > $O/bench_radix_$r.json 2> $O/bench_radix_$r.log
  echo "bench radix=$r rc=$?"
done
python - <<PY
import json
for r in ("default", "1"):
    d = json.loads(open("$O/bench_radix_%s.json" % r).read().strip().splitlines()[-1])
    s = d["secondary"]
    print("radix", r, "headline", round(d["value"], 1), "config4", round(s["config4"]["value"], 1),
          "f2", {k: round(v["value"], 1) for k, v in s["f2"].items() if k.startswith("len")},
          "ref_flags", {k: (round(v["value"], 1), round(v["select_ms_event_timed"], 2)) for k, v in s["rankdata_ref_flags"].items() if k.startswith("len")})
PY
