set -x
cd $GRAFT_REPO_ROOT
python - <<'PY'
import sys, os, json
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import oracle_lib as O, synth_tokenizer
os.makedirs("/tmp/cli", exist_ok=True)
O.synth_write([288,768,6,6,6,32000,256], 1, "/tmp/cli/model.bin")
synth_tokenizer.write("/tmp/cli/tokenizer.bin")
PY
cd /tmp/cli
H=$GRAFT_REPO_ROOT/llama2.ts_amd/host/llama2.mjs
node $H model.bin -t 1.0 -p 0.9 -s 7 -n 24 -i once; echo "rc=$?"
L2_NO_ZERO_COPY=1 node $H model.bin -t 1.0 -p 0.9 -s 7 -n 24 -i once; echo "rc(nozc)=$?"
node --expose-gc -e "
const a=require('$GRAFT_REPO_ROOT/llama2.ts_amd/host/l2_napi.node'); a.open('$GRAFT_REPO_ROOT/llama2.ts_amd/lib/libllama2hip.so');
const c=a.create(new Int32Array([64,176,2,4,4,512,64]),0); a.synthFill(c,1);
const lg=new Float32Array(a.logitsBuffer(c,512),0,512); a.forward(c,1,0,null); console.log(lg[0],lg[1]);
let junk=[]; for(let i=0;i<200000;i++) junk.push({i,p:lg[i%512]}); global.gc(); console.log('gc ok', junk.length);
"; echo "rc(gc)=$?"
