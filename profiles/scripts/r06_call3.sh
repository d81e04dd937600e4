# Round 6, call 3: BASELINE configs[1] as named through the programs against the reference binary (its 5 minutes on the host's
# other socket run beside the GPU work of this call): split rows with their unit's window, stream overlap, the model rounds' kernel
# with chunked rows and per-alignment fields, the drop-in at 5 % with finer marks.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=r06c tools/pin_config.sh configs1 1 10526315 50000 4-12 -- bash profiles/scripts/r06_call3_gpu.sh
