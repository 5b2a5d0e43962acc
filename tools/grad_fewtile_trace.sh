# the gradient launch of mid-size updates under rocprofv3, with variants of kron tuning key 6
for sz in ${SIZES:-1024 1536 2048}; do for k in ${KEYS:-6:0 6:1}; do echo "== $sz key $k"; KRON_KEY=$k bash tools/kron_update_trace.sh $sz f32 2>&1 | grep -E "p3_grad|first start"; done; done
