# one default bench line on this box, reduced to the figures that are compared from box to box
python3 bench.py --no-cpu 2>/dev/null | tail -1 | python3 -c "
import json, sys
d = json.loads(sys.stdin.read())
a = d['also']
print('code %s  headline %.4f ms kernel %.4f frac %.3f (read kernel %.0f GB/s: %.2f of it) | 1024ch %.3f | mixed %.3f | iqdump %.4f ms | ssbmod %.4f ms %.3f | wbfmmod %.3f ms %.3f | realtime p99 %.2f ms' % (
  d['kernel_code_tag'], d['ms_per_step'], d['roofline']['kernel_ms_mean'], d['roofline']['frac'], d['roofline']['measured_stream_read_GBps'], d['roofline']['frac_of_measured_read'],
  a['wbfm_1024x16']['roofline_frac'], a['mixed_256x16']['roofline_frac'], a['wbfm_256x16_iqdump']['ms_per_step'], a['ssbmod_1024x16']['ms_per_step'], a['ssbmod_1024x16']['roofline_frac'],
  a['wbfmmod_1024x16']['ms_per_step'], a['wbfmmod_1024x16']['roofline_frac'], a['realtime_1024x1']['paced_64ms']['p99_ms']))
"
