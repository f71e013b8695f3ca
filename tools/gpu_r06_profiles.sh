#!/bin/bash
# r06 profile set: rocprofv3 kernel stats (three bench modes + the face pass), PMC passes over the chain + conv stacks, PMC of the face's conv
# launches with / without the stream-K band.  Summaries -> gpurun_out/r06_profiles, r06_pmc_face (copied to profiles/ by hand).
bash tools/profile_r06.sh 2>&1 | tail -30
bash tools/pmc_face_r06.sh 2>&1 | tail -30
