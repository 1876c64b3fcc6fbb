#!/bin/bash
# pre-training checkpoint -> fine-tune -> video-level test on the synthetic datasets (tiny config), prints the logs
set -e
OUT=${1:-gpurun_out/ft_demo}
R=$(mktemp -d)          # checkpoints stay out of gpurun_out/ (64 MiB cap); only the logs are copied there
mkdir -p $OUT
COMMON="--dataset synthetic --n_classes 4 --sample_duration 4 --sample_size 32 --model_name r21d_byol --model_depth 1 --n_workers 0 --result_path $R --weight_decay 1e-4"
python main_byol.py $COMMON --batch_size 8 --synthetic_len 32 --task loss_com --loss_weight 0.1 1 1 1 1 --n_epochs 100 --max_steps 1 --learning_rate 0.01 > $R/pretrain.log 2>&1
ls $R/synthetic/loss_com
python main_ft_mp.py $COMMON --batch_size 8 --synthetic_len ${LEN:-64} --task ft_all --pretrained_path $R/synthetic/loss_com/save_100.pth --learning_rate ${LR:-0.02} --n_epochs ${EPOCHS:-6} --lr_patience 1 > $R/ft.log 2>&1
cat $R/synthetic/ft_all/*train*.log $R/synthetic/ft_all/*val*.log
python test.py $COMMON --batch_size 1 --synthetic_len ${LEN:-64} --task test --t_ft_task ft_all > $R/test.log 2>&1
tail -3 $R/test.log
cp $R/*.log $R/synthetic/ft_all/*.log $R/synthetic/test_*.txt $OUT/
rm -rf $R
