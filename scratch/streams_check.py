"""Which hardware queues do the library's side streams get in THIS kind of process?  Runs bench.py's headline step (100 timed steps)
in child processes with RR_SKIP_QUEUES="a,b" -- a / b unused streams created in front of the heavy / very heavy stream -- and prints
the step time of every setting.  Measured on MI355X (round 6): "0,0" 0.585 ms, "1,0" 0.89, "0,1" 0.89, "1,1" 0.585, "0,2" 0.585,
"2,0" 0.85, "3,0" 0.85 -- a host process that creates streams of its own before rr_create may need another setting than the default.
Usage: python scratch/streams_check.py [extra bench.py arguments]"""
import json, os, subprocess, sys
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
for setting in ('0,0', '1,0', '0,1', '1,1', '0,2', '2,0'):
    env = dict(os.environ, RR_SKIP_QUEUES=setting)
    out = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--no-cpu-baseline', '--no-secondary', '--steps', '100', '--warmup', '20'] + sys.argv[1:],
                         env=env, capture_output=True, text=True).stdout
    line = next((l for l in out.splitlines() if l.startswith('{"metric"')), None)
    print('RR_SKIP_QUEUES=%s: %s' % (setting, '%.4f ms per step' % json.loads(line)['ms_per_step'] if line else 'no result'), flush=True)
