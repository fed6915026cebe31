mkdir -p gpurun_out/r4i; O=gpurun_out/r4i
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -30 $O/pytest.log | grep -v "^$" | tail -25
GT_PARITY_REPORT=$PWD/$O/report_bf16.txt timeout 900 python -m pytest tests/test_gpu_at_size.py -m gpu -q -k "bf16" > /dev/null 2>&1
python - <<PY
import re,collections
w=collections.defaultdict(float)
for l in open("$O/report_bf16.txt"):
    m=re.match(r"(\S+)\s+(\S+)\s+rel-rms (\S+)", l)
    if not m: continue
    case,k,err=m.group(1),m.group(2),float(m.group(3))
    kind=k.split(".")[0]
    w[(case,kind)]=max(w[(case,kind)],err)
for k in sorted(w): print(k, "%.2e"%w[k])
PY
GT_PARITY_REPORT=$PWD/$O/report_philox.txt timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "philox_dropout_step" > /dev/null 2>&1; grep -i "census" $O/report_philox.txt
GT_OPT_FUSED=0 timeout 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-other-configs --batch 4 > $O/b4.json 2>/dev/null; GT_D_SPLIT=0 timeout 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-other-configs --batch 4 > $O/b4_nosplit.json 2>/dev/null
GT_D_SPLIT=0 timeout 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-other-configs --batch 8 > $O/b8_nosplit.json 2>/dev/null; timeout 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-other-configs --batch 8 > $O/b8.json 2>/dev/null
for f in b4 b4_nosplit b8 b8_nosplit; do python -c "
import json; d=[json.loads(l) for l in open('$O/$f.json') if l.startswith('{')][-1]; print('$f', d['ms_per_step'])"; done
