cd $GRAFT_REPO_ROOT
for B in 2 4 8 12; do
  for CFG in "-1 -1" "25 26" "30 31" "5 6"; do set -- $CFG
    timeout 120 python tools/profile_driver.py --batch $B --steps 10 --k7 $1 --k3 $2 2>&1 | grep "ms/step"
  done
done
