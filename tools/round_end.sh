# Developer script (GPU box): the round-end sequence -- full GPU suite, default bench line, then the profiles of tools/profile_run.sh
bash tools/final_run.sh && bash tools/profile_run.sh
