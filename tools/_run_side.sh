cd $GRAFT_REPO_ROOT
echo "--- strips, side stream off"
RS_SIDE_STREAM=0 timeout -k 10 300 python tools/strip_balance.py 2>&1 | grep -v "balanced\|amdgpu.ids"
echo "--- strips, side stream on"
timeout -k 10 300 python tools/strip_balance.py 2>&1 | grep -v "balanced\|amdgpu.ids"
