#!/bin/bash
# same-box A/B of the fp32 step: native fp32 MFMA kernels against the fp32x3 mode (six bf16 products per block)
STEPS=${STEPS:-30}
mkdir -p gpurun_out
for tag in x3 f32 x3_noside x3b f32b; do
  case $tag in
    x3|x3b) extra="--precision fp32x3";;
    f32|f32b) extra="";;
    x3_noside) extra="--precision fp32x3 --no_side_stream";;
  esac
  python bench.py --steps $STEPS --warmup 10 --no_cpu_baseline $extra > gpurun_out/ab_$tag.json 2> gpurun_out/ab_$tag.err || tail -5 gpurun_out/ab_$tag.err
done
python - <<'EOF2'
import json
for f in ("x3", "f32", "x3_noside", "x3b", "f32b"):
    try:
        d = json.load(open("gpurun_out/ab_%s.json" % f))
        print(f, d["value"], d["ms_per_step"], d["final_loss"])
        for fam in d.get("roofline_families", []):
            print("   ", fam["family"], fam["launches_per_step"], fam["avg_us"], fam["ms_per_step"], fam["achieved"])
    except Exception as e:
        print(f, "ERR", e)
EOF2
