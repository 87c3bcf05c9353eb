for r in 1 2 3; do
  for v in default latefrags; do
    if [ $v = default ]; then unset CODD_KNN_LIB; else export CODD_KNN_LIB=$PWD/codd_query_engine_amd/csrc/libcodd_knn_$v.so; fi
    echo -n "$v "; timeout -k 10 200 python scripts/ab_i8.py ${1:-4000000} 2>/dev/null | sed 's/.*"v1": {\([^}]*\)}.*/\1/'
  done
done
