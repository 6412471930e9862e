#!/bin/bash
# instruction mix of k_integrate<2>'s voxel-update path (first voxel load .. the block's barrier) from build/engine.s
A=ra-slam_amd/csrc/build/engine.s
S=$(grep -n "^_ZN6ratsdf11k_integrateILi2EEEvNS_9IntegArgsENS_11FrameParamsEPU3AS4KNS_9EngineDevEjjjjNS_7CandJobE:" $A | cut -d: -f1)
E=$(awk -v s=$S 'NR>s && /\.amdhsa_next_free_vgpr/{print NR; exit}' $A)
sed -n "${S},${E}p" $A > /tmp/kint2.s
sed -n "$E,$((E+1))p" $A
grep -n "private_segment_fixed_size" /tmp/kint2.s
# hot path: from the label before the first 'global_load_dwordx2 v[..], v[..], off' with plain addressing inside the loop to the first s_barrier after it
L0=$(grep -n "global_load_dwordx2 v\[[0-9:]*\], v\[[0-9:]*\], off$" /tmp/kint2.s | head -1 | cut -d: -f1)
L1=$(awk -v s=$L0 'NR>s && /s_barrier/{print NR; exit}' /tmp/kint2.s)
L0=$((L0-20))
sed -n "${L0},${L1}p" /tmp/kint2.s > /tmp/hot.s
echo "lines $L0-$L1: VALU $(grep -c '^\s*v_' /tmp/hot.s) (incl. rare IEEE-div block $(sed -n '/v_div_scale_f32/,/^\.LBB/p' /tmp/hot.s | grep -c '^\s*v_')) pk $(grep -c '^\s*v_pk_' /tmp/hot.s) s_nop $(grep -c 's_nop' /tmp/hot.s) readlane $(grep -c 'v_readlane' /tmp/hot.s) ds $(grep -c '^\s*ds_' /tmp/hot.s) salu $(grep -c '^\s*s_' /tmp/hot.s) smem $(grep -c '^\s*s_load' /tmp/hot.s)"
