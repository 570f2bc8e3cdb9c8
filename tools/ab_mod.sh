# SSB modulator (BASELINE config 5) on every variant library
for v in $(ls hackrfdiags_amd/lib/variants); do echo -n "$v "; HRFD_LIB=hackrfdiags_amd/lib/variants/$v/libhrfd.so python bench.py --workload ssbmod --channels 1024 --no-cpu 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'])"; done
