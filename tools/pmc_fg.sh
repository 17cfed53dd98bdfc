cd /tmp && export TMPDIR=/tmp
for mode in wide narrow; do
  if [ $mode = narrow ]; then export GTNX_FIXED_GRAD_NARROW=1; else unset GTNX_FIXED_GRAD_NARROW; fi
  rm -rf /tmp/pf_$mode
  timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pf_$mode -- python /root/repo/tools/bench_c4.py --no-cpu-baseline --steps 1 > /dev/null 2>&1
  F=$(find /tmp/pf_$mode -name "*counter_collection.csv" | head -1)
  python - "$F" $mode <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot={}
for r in rows:
    n=r["Kernel_Name"]
    if "fixed_grad" in n or "z_chain" in n or "mfma_step" in n:
        k=n.replace("(anonymous namespace)::","").split("(")[0][-45:]
        tot.setdefault(k,[0,0.0]); tot[k][0]+=1; tot[k][1]+=float(r["Counter_Value"])
for k,v in tot.items(): print(sys.argv[2], k, "launches", v[0], "FETCH_SIZE per launch (KiB)", round(v[1]/v[0]))
PY
done
