set -e
for w in c2 c1 c2v c3 c4 c5; do
  timeout -k 10 1000 bash tools/profile_round.sh r02b_$w $w > gpurun_out/prof_r02b_$w.log 2>&1 || { echo "profile $w failed"; tail -5 gpurun_out/prof_r02b_$w.log; exit 1; }
  echo "$w done"
done
