cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python tools/dev/copy_probe.py 2>&1 | tail -1
PROBE_TAG=blit_engine_2 GPU_BLIT_ENGINE_TYPE=2 python tools/dev/copy_probe.py 2>&1 | tail -1
PROBE_TAG=force_blit_0 GPU_FORCE_BLIT_COPY_SIZE=0 python tools/dev/copy_probe.py 2>&1 | tail -1
PROBE_TAG=limit_blit_wg_8 DEBUG_CLR_LIMIT_BLIT_WG=8 python tools/dev/copy_probe.py 2>&1 | tail -1
PROBE_TAG=sdma_off HSA_ENABLE_SDMA=0 python tools/dev/copy_probe.py 2>&1 | tail -1
