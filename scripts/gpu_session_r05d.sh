#!/bin/bash
# Round 5, session d: the BatchNorm backward applied by the input-gradient GEMM (ptt_rows_gemm_bnbwd_fused_f32), FPS with the
# candidate coordinates prefetched under the wave reduction: parity, then the training step and one tracklet frame
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r05d; mkdir -p $O
timeout 1200 python -m pytest tests/test_round5_gpu.py tests/test_point_ops_gpu.py tests/test_train_gpu.py tests/test_train_config3_gpu.py tests/test_golden_gpu.py tests/test_tracking_gpu.py tests/test_hot_path_gpu.py -x -q -m gpu > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
tail -12 $O/pytest.log | cut -c1-300
for m in 1 0 1 0; do
  PTT_FUSED_BN_BWD=$m timeout 300 python bench.py --workload train --steps 20 --warmup 5 --sustain 2 --no-cpu-baseline 2> $O/train_f$m.err | tee -a $O/train_modes.jsonl | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fused $m', d['ms_per_step'], d['sustained'])"
done
timeout 300 python scripts/tracklet_b1_profile.py 2>&1 | grep -v amdgpu.ids | tee $O/b1.log
timeout 300 python scripts/fps_sweep.py > $O/fps_sweep.log 2>&1; tail -30 $O/fps_sweep.log
