cd $GRAFT_REPO_ROOT
for v in azplugins_amd/libazp.so tools/libazp_abl4.so azplugins_amd/libazp.so tools/libazp_abl4.so; do
  echo "lib: $v"
  AZP_LIB_PATH=$v python3 bench.py --no-cpu-baseline --no-verify --steps 200 --warmup 20 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('  kernel_ms %.4f'%d['roofline']['kernel_ms'], ' by step', ' '.join('%.4f'%x['kernel_ms'] for x in d['config']['kernel_ms_by_cycle_step']), ' whole rows %.4f'%d['config']['kernel_ms_other_states'].get('whole_rows_no_displacement_information_ms',0))
"
done
