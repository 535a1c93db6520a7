timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "teacher_forced_config4 or landscape or panel or trajectory_is_bit or ipc_row" 2>&1 | tail -3
for r in 1 2; do for v in 0 1; do
  MACHIP_PANEL_OPS=$v timeout 300 python bench.py --no-cpu --no-pmc --no-warm --no-same-node 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('panel_ops $v', round(d['value'],1), d['lanczos_steps_per_iter'], round(d['ms_per_step'],4))"
done; done
