# Round 6, call 6: BASELINE configs[4] at a tenth (10 M reads x 500 k transcripts, ~40 alignments per read) through the programs with
# --lean-device against the reference binary (its run on the last socket beside:) the -b pass at 10 % of configs[2] with BAM input
# against the reference, and configs[2] at full size with finer host marks.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
df -h /tmp | tail -1; free -g | head -2
DROPIN_FLAGS="--lean-device" TAG=r06f tools/pin_config.sh configs4_tenth 1 10526315 500000 32-64 -- bash profiles/scripts/r06_call6_gpu.sh
