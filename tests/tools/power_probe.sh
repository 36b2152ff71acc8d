# Samples the GPU's power draw and shader clock (rocm-smi, read-only) while bench.py runs: is the step power-limited?
# Usage on the GPU box: bash tests/tools/power_probe.sh "<bench args>"   -> prints the bench line's value and min / median / max of the samples
ARGS="$1"
python bench.py --steps 2500 --warmup 10 --no_cpu_baseline --no_native_leg --no_bf16_leg --prof_kind 0 $ARGS > /tmp/pp_bench.json 2> /tmp/pp_bench.err &
BP=$!
sleep 6      # model build; samples taken while the GPU idles are dropped below (sclk < 500 MHz)
: > /tmp/pp_samples.txt
for i in $(seq 1 70); do
  rocm-smi --showpower --showclocks --json 2>/dev/null >> /tmp/pp_samples.txt
  echo >> /tmp/pp_samples.txt
  sleep 0.3
done
wait $BP
python - "$ARGS" <<'EOF2'
import json, sys, statistics as st
pw, sc, mc = [], [], []
for line in open('/tmp/pp_samples.txt'):
    line = line.strip()
    if not line.startswith('{'):
        continue
    try:
        d = json.loads(line)
    except Exception:
        continue
    c = d.get('card0') or next(iter(d.values()))
    for k, v in c.items():
        kl = k.lower()
        try:
            if 'power' in kl and 'w' in kl: pw.append(float(str(v).split()[0]))
            elif 'sclk' in kl and 'level' not in kl or kl.startswith('sclk'): sc.append(float(str(v).strip('()').replace('Mhz', '').replace('MHz', '').split()[0]))
        except Exception:
            pass
keep = [i for i in range(min(len(pw), len(sc))) if sc[i] >= 500]
pw = [pw[i] for i in keep]; sc = [sc[i] for i in keep]
b = json.load(open('/tmp/pp_bench.json'))
def s(x): return 'n/a' if not x else '%.0f / %.0f / %.0f' % (min(x), st.median(x), max(x))
print('%-40s %8.1f samples/s %7.3f ms | power W (min / med / max) %s | sclk MHz %s | %d samples' % (sys.argv[1] or '(default)', b['value'], b['ms_per_step'], s(pw), s(sc), len(pw)))
EOF2
