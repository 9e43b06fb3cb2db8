for l in tree mt5 mt7 mt8 pw6 pw8 tree; do
if [ $l = tree ]; then unset METHEOR_HIP_LIB; else export METHEOR_HIP_LIB=$PWD/abx/lib$l.so; fi
timeout 300 python bench.py --legs all7 2>/dev/null | python -c "
import json,sys
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1])['all7']
print('$l', j['per_pass_ms_one_sync_each'], 'prepared', j['prepared_batches']['per_pass_ms_one_sync_each'])"
done
