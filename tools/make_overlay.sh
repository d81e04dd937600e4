#!/bin/bash
# An RSEM installation directory in which some programs are the drop-ins of this repo.
#   tools/make_overlay.sh <rsem_dir> <out_dir> [program ... | --none]      default programs: all four drop-ins
# rsem-calculate-expression puts ITS OWN directory first in PATH (rsem-calculate-expression:12, FindBin::RealBin, symlinks
# resolved) and runs bare program names, so a drop-in has to sit next to the driver: <out_dir> gets COPIES of the Perl
# scripts (a symlinked script would resolve back to <rsem_dir>), symlinks to everything else in <rsem_dir>, and symlinks to
# rsem_amd/bin/<program> for the named programs (their rpath $ORIGIN/.. finds librsem_hip.so through the resolved path).
set -e
src=$(cd "$1" && pwd); out=$2; shift 2
here=$(cd "$(dirname "$0")/.." && pwd)
progs=("$@"); [ ${#progs[@]} -eq 0 ] && progs=(rsem-parse-alignments rsem-run-em rsem-run-gibbs rsem-calculate-credibility-intervals)
[ "${progs[0]}" == "--none" ] && progs=()
mkdir -p "$out"
for f in "$src"/*; do
  b=$(basename "$f")
  case "$b" in
    *.pm|rsem-calculate-expression|rsem-prepare-reference) cp "$f" "$out/$b" ;;
    obj|gen|*.a) ;;
    *) ln -sfn "$f" "$out/$b" ;;
  esac
done
for p in "${progs[@]}"; do
  [ -x "$here/rsem_amd/bin/$p" ] || { echo "make_overlay: $here/rsem_amd/bin/$p is missing (python -m rsem_amd.build)" >&2; exit 1; }
  ln -sfn "$here/rsem_amd/bin/$p" "$out/$p"
done
echo "$out: ${progs[*]} from $here/rsem_amd/bin, the rest from $src"
