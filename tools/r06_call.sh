# The script of one `gpurun` call of round 6 (rewritten for every call; the outputs of each are under profiles/r06*).
# Last state: the whole-state check of the final build.
set -u
export TMPDIR=/tmp
TAG=${TAG:-r06s} SHORT=1 bash tools/round_check.sh
