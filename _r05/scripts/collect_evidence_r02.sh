set -x
O=gpurun_out/r02f; mkdir -p $O
export WCMC_DEBUG_LIB=1
WCMC_HALO64=0 WCMC_DEBUG_ABLATE=64 timeout 200 python3 scripts/timeline_halo.py 124 116 100 2>&1 | grep -v amdgpu.ids > $O/halo8x16_timeline.txt
WCMC_DEBUG_ABLATE=64 timeout 200 python3 scripts/timeline_halo.py 124 2>&1 | grep -v amdgpu.ids > $O/halo64_timeline.txt
# (the 12x16-tile instance has no stamp build: the 104x104 layer on 16x16 tiles)
(echo "# WCMC_HALO64_PT3=0"; WCMC_HALO64_PT3=0 WCMC_DEBUG_ABLATE=64 timeout 200 python3 scripts/timeline_halo.py 108 2>&1 | grep -v amdgpu.ids) >> $O/halo64_timeline.txt
WCMC_HALO64=0 timeout 300 python3 scripts/time_halo_abl.py 124 0 1 2 4 8 16 32 10 26 2>&1 | grep -v amdgpu.ids > $O/halo8x16_ablations.txt
timeout 300 python3 scripts/time_halo_abl.py 124 0 1 2 4 8 10 14 32 46 2>&1 | grep -v amdgpu.ids > $O/halo64_ablations.txt
(echo "# conv_wgrad_rows8_bf16x3_kernel (shipped: priority hand-over at iteration 8 of 14)"; WCMC_DEBUG_ABLATE=16 timeout 200 python3 scripts/timeline_wgrad.py 124 108 2>&1 | grep -v amdgpu.ids;
 echo "# ... WCMC_WGRAD_ROWS8_PRIO=0 (no hand-over)"; WCMC_WGRAD_ROWS8_PRIO=0 WCMC_DEBUG_ABLATE=16 timeout 200 python3 scripts/timeline_wgrad.py 124 2>&1 | grep -v amdgpu.ids;
 echo "# WCMC_WGRAD_ROWS8=0: conv_wgrad_rows_bf16x3_kernel<5, 7, 7> (seven waves)"; WCMC_WGRAD_ROWS8=0 WCMC_DEBUG_ABLATE=16 timeout 200 python3 scripts/timeline_wgrad.py 124 108 2>&1 | grep -v amdgpu.ids) > $O/wgrad_rows_timeline.txt
(echo "# WCMC_WGRAD_ROWS8=0: the ablation instances are those of the seven-wave kernel"; WCMC_WGRAD_ROWS8=0 timeout 300 python3 scripts/time_wgrad_abl.py 124 2>&1 | grep -v amdgpu.ids) > $O/wgrad_rows_ablations.txt
(echo "# conv_wgrad_rows8_bf16x3_kernel: 1 no MFMA, 2 no fills after the first, 3 both, 8 no fragment waits, 32 no fragment reads, 34 = 32 + 2, 35 = all"; timeout 300 python3 scripts/time_wgrad_abl.py 124 0 1 2 3 8 32 34 35 2>&1 | grep -v amdgpu.ids) > $O/wgrad_rows8_ablations.txt
WCMC_DEBUG_LIB= PRIOS=0,8 XES=0,1 timeout 300 python3 scripts/time_wgrad_rows8.py 2>&1 | grep -v amdgpu.ids > $O/wgrad_rows8.txt
unset WCMC_DEBUG_LIB
timeout 250 python3 scripts/time_conv_layers.py 2>&1 | grep -v amdgpu.ids > $O/conv_layers.txt
WCMC_HALO64=0 timeout 250 python3 scripts/time_conv_layers.py 2>&1 | grep -v amdgpu.ids > $O/conv_layers_halo8x16.txt
WCMC_HALO64_PT3=0 timeout 250 python3 scripts/time_conv_layers.py 2>&1 | grep -v amdgpu.ids > $O/conv_layers_pt4_only.txt
timeout 200 python3 scripts/time_wgrad.py 2>&1 | grep -v amdgpu.ids > $O/wgrad_layers.txt
timeout 300 python3 scripts/time_wgrad_1x1.py 2>&1 | grep -v amdgpu.ids > $O/wgrad_1x1.txt
timeout 250 python3 scripts/time_wgrad_unet.py 2>&1 | grep -v amdgpu.ids > $O/wgrad_unet.txt
WCMC_WGRAD_ROWS_3X3=0 timeout 250 python3 scripts/time_wgrad_unet.py 2>&1 | grep -v amdgpu.ids > $O/wgrad_unet_onetap.txt
timeout 250 python3 scripts/time_unet_layers.py 2>&1 | grep -v amdgpu.ids > $O/unet_layers.txt
ls -la $O
