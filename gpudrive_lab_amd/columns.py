"""Column contract of the exported tensors: what the reference's Python decoders index
(gpudrive/datatypes/observation.py, roadgraph.py, info.py, metadata.py, control.py, trajectory.py) and the
structs that define it (reference src/types.hpp:188-441).  Pinned by golden vectors produced with those
decoders (tests/golden/make_columns_golden.py) and by cross-tensor invariants (tests/test_columns.py)."""

SELF_OBS = dict(speed=0, vehicle_length=1, vehicle_width=2, vehicle_height=3, rel_goal_x=4, rel_goal_y=5,
                is_collided=6, id=7)                                    # types.hpp:188-208, [W, A, 8]
ABS_OBS = dict(pos_x=0, pos_y=1, pos_z=2, rotation_as_quaternion=slice(3, 7), rotation_angle=7, goal_x=8,
               goal_y=9, vehicle_length=10, vehicle_width=11, vehicle_height=12, id=13)  # :395-406, [W, A, 14]
PARTNER_OBS = dict(speed=0, rel_pos_x=1, rel_pos_y=2, orientation=3, vehicle_length=4, vehicle_width=5,
                   vehicle_height=6, agent_type=7, ids=8)               # :236-275, [W, A, A-1, 9]
ROAD_ROW = dict(x=0, y=1, segment_length=2, segment_width=3, segment_height=4, orientation=5, type=6, id=7,
                vbd_type=8)                                             # :210-294, [W, A, 200, 9] and [W, 10000, 9]
INFO = dict(off_road=0, collided=slice(1, 3), goal_achieved=3, agent_type=4)   # :379-393, [W, A, 5] int32
METADATA = dict(is_sdc=0, objects_of_interest=1, tracks_to_predict=2, difficulty=3)   # :426-441, [W, A, 4] int32
RESPONSE_TYPE = dict(moving=0, kinematic=1, static=2)                   # values of response_type_tensor
# the reference multiplies these by madrona_gpudrive.vehicleScale when decoding (observation.py:15-16, 34-35, 182-183)
SCALED_BY_AGENT_SCALE = {"SELF_OBS": ("vehicle_length", "vehicle_width"), "ABS_OBS": ("vehicle_length", "vehicle_width"),
                         "PARTNER_OBS": ("vehicle_length", "vehicle_width")}
