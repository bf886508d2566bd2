#!/bin/bash
# reference run_scripts/one_step_nerf_only_train.sh on the synthetic scenes
name=one_step_nerf_only
python scripts/train_joint.py --exp cfg/exp/synthetic/s00.yml --exp_name $name --project_name $name --nerf_train_epoch 60 --joint_train_epoch 0
