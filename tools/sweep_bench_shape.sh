for cfg in "20 5 4" "20 4 4" "20 3 4" "20 2 4" "20 7 3" "20 10 2" "20 4 5" "40 5 4" "40 8 4" "40 10 4" "60 5 4" "20 5 4"; do
  set -- $cfg
  v=$(timeout 120 python bench.py --steps $1 --coalesce $2 --in-flight $3 --warmup 2 --no-cpu-baseline --no-alt --no-a2m --no-novae --no-clip --no-rocprof 2>/dev/null | python -c "import sys,json; b=json.loads(sys.stdin.read()); print(b['value'], b['ms_per_step'])")
  echo "steps=$1 coalesce=$2 inflight=$3 -> $v"
done
