#!/bin/bash
# Builds A/B variants of the engine, one .so each, for tools/ab_f64.py and tools/ab_basket.py (both time every
# tools/ab_*.so they find, interleaved in one process on one device).  The tree itself carries no experiment switches
# (round 3 removed the decided MC_AB_* branches): a variant is the working tree, a patch on top of it, or another revision.
#   tools/build_ab.sh NAME=SPEC ...
#     SPEC  (empty)        montecarlocuda_amd/csrc as it is
#           -Dx,-Dy        ... with extra compiler flags (commas for spaces)
#           @file.patch    ... with the patch applied to a scratch copy (patch -p1 from the repo root)
#           rev:<git-rev>  csrc of that revision (git archive)
#   e.g.  tools/build_ab.sh 0_stream_v1=rev:HEAD~1 1_stream_v2=
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
F="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=fast -w"
rm -f "$ROOT"/tools/ab_*.so
build() {   # name, source dir, extra flags
    local name=$1 src=$2 flags=$3 tmp
    tmp=$(mktemp -d /tmp/mcab.XXXXXX)
    ( cd "$src" && /opt/rocm/bin/hipcc $F $flags -c -o "$tmp/mc_api.o" mc_api.hip &&
      if [ -f mc_hostmath.c ]; then gcc -O2 -std=gnu11 -ffp-contract=off -fPIC -c -o "$tmp/mc_hostmath.o" mc_hostmath.c; fi &&
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o "$ROOT/tools/ab_${name}.so" "$tmp"/*.o )
    rm -rf "$tmp"
}
n=0
for spec in "$@"; do
    name="${spec%%=*}"; what="${spec#*=}"
    case "$what" in
        rev:*)  d=$(mktemp -d /tmp/mcabsrc.XXXXXX); git -C "$ROOT" archive "${what#rev:}" montecarlocuda_amd/csrc include | tar -x -C "$d"
                build "$name" "$d/montecarlocuda_amd/csrc" "" & ;;
        @*)     d=$(mktemp -d /tmp/mcabsrc.XXXXXX); mkdir -p "$d/montecarlocuda_amd"; cp -r "$ROOT/montecarlocuda_amd/csrc" "$d/montecarlocuda_amd/"; cp -r "$ROOT/include" "$d/"
                ( cd "$d" && patch -p1 -s < "$ROOT/${what#@}" ); build "$name" "$d/montecarlocuda_amd/csrc" "" & ;;
        *)      build "$name" "$ROOT/montecarlocuda_amd/csrc" "${what//,/ }" & ;;
    esac
    n=$((n + 1)); if [ $((n % 3)) = 0 ]; then wait; fi
done
wait
ls -la "$ROOT"/tools/ab_*.so
