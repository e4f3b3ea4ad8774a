#!/bin/bash
# Stage the Python reference under the git-ignored oracle/_ref/ so that it travels to the
# GPU box as test infrastructure (see oracle/stage_reference.py).  Build container only.
exec python "$(dirname "$0")/../oracle/stage_reference.py" "$@"
