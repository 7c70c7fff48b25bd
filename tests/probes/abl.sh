for L in "" $(ls pathfinder.jl_amd/build/abl/libpfmi_abl${ABL:-?}.so); do
  if [ -n "$L" ]; then export PFMI_LIB_PATH=$PWD/$L; fi
  python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; l=json.loads(sys.stdin.read()); print('$L', l['ms_per_step'], l['stages_ms']['history'], l['stages_ms']['fit'])"
done
