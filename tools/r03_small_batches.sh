# batches of small objects on the launch-per-step route (one handle, device-resident): "<n>:<objects>:<m>"
S="512:128:19 512:64:19 512:256:19 256:256:19 384:128:64 640:128:19 768:64:64 1024:64:64 1024:128:19 1280:32:64 1536:32:64 1920:16:64 900:40:19 130:500:19"
timeout -k 5 300 python3 tools/batch_time.py "$S" CGP_NO_DAG_BATCH=1 | sed 's/dag\(.*\)steps.*/launch per step\1/'
