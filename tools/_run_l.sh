cd /root/repo
export TMPDIR=/tmp
python tools/sharded_bench.py --frames 4 --steps 200
python tools/sharded_bench.py --frames 3 --steps 200
echo "== bench --force-dist"
MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 python bench.py --force-dist --steps 200 --warmup 10 --cpu-sample 0 --single-frames 0 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('   value %.1f M  ms/step %.3f scenes %d' % (d['value']/1e6, d['ms_per_step'], d['config']['distinct_scenes']))"
echo "== bench default 4 scenes"
python bench.py --steps 200 --warmup 10 --cpu-sample 0 --single-frames 0 --scenes 4 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('   value %.1f M  ms/step %.3f scenes %d' % (d['value']/1e6, d['ms_per_step'], d['config']['distinct_scenes']))"
