#!/bin/bash
# every variant of tools/drift_ab.py -> gpurun_out/drift_ab.jsonl (builds: see the header of drift_ab.py; csrc/build.sh with
# ZEGGS_DEFS / ZEGGS_OUT / ZEGGS_BUILD_DIR)
Z=ubisoft-laforge-zeroeggs_amd/zeggs
O=gpurun_out/drift_ab.jsonl
mkdir -p gpurun_out; rm -f $O
run() { tag=$1; lib=$2; opts=$3; shift 3; ZEGGS_LIB=$lib ZEGGS_OPTIONS=$opts timeout 300 python tools/drift_ab.py $tag $O "$@" 2>&1 | tail -1 | cut -c1-400; }
run persistent            $Z/libzeggs_hip.so ""
run persistent_perturbed  $Z/libzeggs_hip.so "" perturb
run persistent_exact_gates  $Z/libzeggs_exg.so ""
run persistent_exact_sincos $Z/libzeggs_exs.so ""
run persistent_exact_both   $Z/libzeggs_exb.so ""
run stage_launches        $Z/libzeggs_hip.so "persistent=0"
run stage_exact_both      $Z/libzeggs_exb.so "persistent=0"
run generic_unfolded      $Z/libzeggs_hip.so "decoder_fast=0"
run generic_exact_both    $Z/libzeggs_exb.so "decoder_fast=0"
run generic_exact_both_perturbed $Z/libzeggs_exb.so "decoder_fast=0" perturb
# round-4 diagnostics: the root integration of the generic path in float64 / the whole library without
# FMA contraction
run generic_exact_root_fp64 $Z/libzeggs_r64.so "decoder_fast=0"
run generic_exact_no_fma    $Z/libzeggs_nofma.so "decoder_fast=0"
