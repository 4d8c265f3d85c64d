for e in 0 1 0 1; do
  ESVO_DEV_SWITCHES=1 ESVO_LOWLAT=$e python tools/band_share_probe.py dsec640x480 8 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin); g=d['G8']
print('LOWLAT=$e', 'full', round(d['full_tick_ms_synchronised']['total'],4), 'share', {k: round(v,4) for k,v in g['rank_share_ms'].items()}, {k: round(v,4) for k,v in g['stages_ms_mean'].items()}, 'speedup', round(g['projected_speedup_compute_only'],3), 'rep_frac', round(d['replicated_frac_of_rank_share_at_8'],4), g['map_equal_to_one_gpu'])"
done
