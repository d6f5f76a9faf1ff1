#!/bin/bash
# Copy one profile run's summaries from the scratch directory the GPU box merged back (gpurun_out/prof_<tag>/) into the
# tracked profiles/ directory:  tools/collect_profiles.sh r02_z
set -e
tag=$1
R=$(cd "$(dirname "$0")/.." && pwd)
src=$R/gpurun_out/prof_$tag
for f in "$src"/${tag}_*; do
  b=$(basename "$f")
  case "$b" in
    *.md) sed -e "s#(gpurun_out/prof_${tag}/#(${tag}: #g" -e "s#$src/##g" "$f" > "$R/profiles/$b" ;;
    *) cp "$f" "$R/profiles/$b" ;;
  esac
done
ls "$R/profiles" | grep "^${tag}_" | tr '\n' ' '; echo
