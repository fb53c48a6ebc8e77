#!/bin/bash
# like ab.sh for a python tool that honours BMO_ENGINE_LIB: tools/ab_libs.sh "<python tool and args>" lib1 lib2 ...  (REPS rounds, interleaved)
cmd=$1; shift
for rep in $(seq 1 ${REPS:-2}); do for lib in "$@"; do BMO_ENGINE_LIB=$PWD/$lib python $cmd || exit 1; done; done
