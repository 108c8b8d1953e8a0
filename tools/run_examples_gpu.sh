#!/bin/bash
# Smoke-run the example programs on the GPU with tiny budgets.  Each step has
# its own time limit; the script stops at the first failure.
set -o pipefail
cd "$(dirname "$0")/.."
OUT=gpurun_out/examples_gpu
mkdir -p $OUT
run() {
  name=$1; shift
  echo "== $name: $*"
  timeout -k 10 170 "$@" > $OUT/$name.log 2>&1
  rc=$?
  tail -2 $OUT/$name.log
  if [ $rc -ne 0 ]; then echo "FAILED $name rc=$rc"; exit $rc; fi
}
run cnn_mnist python examples/cnn/train_cnn.py cnn mnist -m 1 --max-batches 20
run cnn_mnist_fp16 python examples/cnn/train_cnn.py cnn mnist -p float16 -m 1 --max-batches 20
run resnet_cifar python examples/cnn/train_cnn.py resnet cifar10 -m 1 -b 16 --max-batches 4
run alexnet_cifar python examples/cnn/train_cnn.py alexnet cifar10 -m 1 -b 8 --max-batches 3
run xception_cifar python examples/cnn/train_cnn.py xceptionnet cifar10 -m 1 -b 4 --max-batches 2
run mlp_model python examples/mlp/model.py -m 101
run mlp_native python examples/mlp/native.py -m 101
run benchmark python examples/cnn/benchmark.py --iters 10
run char_rnn python examples/rnn/char_rnn.py -m 1 -b 16 -l 32 -d 64
run imdb python examples/rnn/imdb_train.py -m 1 -b 32
run qabot python examples/qabot/qabot_train.py -m 1 -b 16
run gan python examples/gan/vanilla.py --use_gpu --iterations 40 --interval 20
run lsgan python examples/gan/lsgan.py --use_gpu --iterations 40 --interval 20
run rbm python examples/rbm/train.py --use_gpu -m 1 --hdim 64
run onnx_train python examples/onnx/training/train.py --self-export -m 1 -b 8 --max-batches 3
run onnx_selftest python examples/onnx/text_models.py self-test
run largedata python examples/largedataset_cnn/train_largedata.py cnn --synthetic -m 1 -b 16
run hfl python examples/hfl/run_local.py --num_clients 2 -m 2 --port 15411
run caffe_lenet python examples/caffe/train.py --iters 100 --save /tmp/singa_caffe_lenet.caffemodel
run caffe_lenet_restore python examples/caffe/train.py --iters 10 --weights /tmp/singa_caffe_lenet.caffemodel
run cpp_cnn singa_amd/bin/singa_cnn -epoch 1
run cpp_cnn_opencl singa_amd/bin/singa_cnn -epoch 1 -opencl
run cpp_cnn_parallel singa_amd/bin/singa_cnn_parallel -epoch 1 -replicas 1
run cpp_vgg_parallel singa_amd/bin/singa_vgg_parallel -epoch 3 -replicas 1 -dropout 0.5
run cpp_alexnet singa_amd/bin/singa_alexnet -synthetic -data /tmp/singa_alexnet_gpu
run autograd_mnist python examples/cnn/autograd/mnist_cnn.py -m 1 --max-batches 30
run autograd_resnet_cifar python examples/cnn/autograd/resnet_cifar10.py -m 1 --max-batches 20
run autograd_xception python examples/cnn/autograd/xceptionnet.py -n 10
run autograd_resnet_dist python -m torch.distributed.run --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29641 examples/cnn/autograd/resnet_dist.py -n 20
run autograd_sparsification python -m torch.distributed.run --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29642 examples/cnn/autograd/sparsification_mnist.py -m 1 --max-batches 20
for opt in plain half partialUpdate sparseTopK sparseThreshold; do
  run cifar_dist_$opt env SINGA_DIST_FORCE=1 MASTER_PORT=29651 python examples/cifar_distributed_cnn/train_mpi.py resnet cifar10 -b 32 -d $opt
done
run cifar_dist_half_fp16 env SINGA_DIST_FORCE=1 MASTER_PORT=29652 python examples/cifar_distributed_cnn/train_mpi.py resnet cifar10 -b 32 -d half -p float16
run cnn_multiprocess_gloo env SINGA_DIST_BACKEND=gloo python examples/cnn/train_multiprocess.py cnn mnist -m 1 --max-batches 5 -w 2
run malaria python examples/malaria_cnn/train_cnn.py cnn malaria --synthetic -m 3 -b 32
run malaria_mlp python examples/malaria_cnn/train_cnn.py mlp malaria --synthetic -m 2 -b 32
run bloodmnist python examples/demos/Classification/BloodMnist/ClassDemo.py -m 1 --max-batches 3
run lime python examples/singa_easy/lime.py
run onnx_zoo_resnet18 python examples/onnx/resnet18.py --repeat 10
run onnx_zoo_mobilenet python examples/onnx/mobilenet.py --repeat 10
run onnx_zoo_yolo python examples/onnx/tiny_yolov2.py
run onnx_zoo_sr python examples/onnx/superresolution.py
run onnx_bert_squad python examples/onnx/bert/bert-squad.py
run onnx_gpt2 python examples/onnx/gpt2/gpt2.py --length 5
run onnx_roberta python examples/onnx/ro_bert_a/ro_bert_a.py
run model_selection env PYTHONPATH=examples/model_selection:. python -m trails.run_ms --budget 15
run model_selection_scores python examples/model_selection/exps/score_correlation.py --archs 4 --epochs 1 --rows 256 --gpu
run model_selection_slicing python examples/model_selection/slicing/baseline_int.py --rows 512 --epochs 1 --repeat 3
echo ALL-EXAMPLES-OK
