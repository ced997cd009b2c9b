# round-3 closing run: the whole GPU suite, then every judged artefact (profiles/refresh.sh); through gpurun from the repo root
cd $GRAFT_REPO_ROOT
o=gpurun_out/r03
mkdir -p $o
timeout -k 10 700 python -m pytest tests -q -m gpu > $o/t_final.log 2>&1; echo "pytest rc=$?"; tail -4 $o/t_final.log
bash profiles/refresh.sh && echo refresh-ok
