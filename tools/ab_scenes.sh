#!/bin/bash
# tools/ab_scenes.sh <rays> lib1 lib2 ... : per-scene kernel times for several engine builds
rays=$1; shift
for lib in "$@"; do BMO_ENGINE_LIB=$PWD/$lib python tools/scene_times.py $rays || exit 1; done
