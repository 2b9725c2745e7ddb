#!/bin/bash
# Half of tools/r03_profile_all.sh (a gpurun call is limited to 20 minutes):  tools/r03_profile_half.sh <1|2> <suffix>
H=$1; SFX=$2
cd ${GRAFT_REPO_ROOT:-.}
if [ "$H" = 1 ]; then
tools/run_profile.sh ab30$SFX abmpc > gpurun_out/prof_ab30$SFX.log 2>&1
EEPACC_PROFILE_NO_DRIVER=1 tools/run_profile.sh ab30d$SFX abmpc --steps 20 --warmup 5 > gpurun_out/prof_ab30d$SFX.log 2>&1
EEPACC_PROFILE_NO_DRIVER=1 tools/run_profile.sh bl30d$SFX blmpc --steps 20 --warmup 5 > gpurun_out/prof_bl30d$SFX.log 2>&1
else
tools/run_profile.sh fb30$SFX fbmpc > gpurun_out/prof_fb30$SFX.log 2>&1
EEPACC_PROFILE_NO_DRIVER=1 tools/run_profile.sh fb30d$SFX fbmpc --steps 20 --warmup 5 > gpurun_out/prof_fb30d$SFX.log 2>&1
EEPACC_PROFILE_NO_DRIVER=1 tools/run_profile.sh ab60d$SFX abmpc --horizon 60 --batch 8192 --steps 20 --warmup 5 > gpurun_out/prof_ab60d$SFX.log 2>&1
fi
for f in gpurun_out/prof_*$SFX.log; do echo $f; tail -n 3 $f; done
