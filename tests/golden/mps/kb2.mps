NAME          KB2
ROWS
 N  FAT7..J.
 E  BAL...BW
 E  BHC...BW
 E  BLC...BW
 E  BLV...BW
 E  BN4...BW
 E  BP8...BW
 E  BTO...BW
 E  B3E...BW
 E  B3P...BW
 E  B3R...BW
 E  B3T...BW
 E  B3E.VOBW
 E  B3P.VOBW
 E  B3R.VOBW
 G  HMH.3EBW
 G  HML.3EBW
 G  HMM.3EBW
 G  HRH.3EBW
 G  HRL.3EBW
 G  HRM.3EBW
 G  HMH.3RBW
 G  HML.3RBW
 G  HMM.3RBW
 G  HRH.3RBW
 G  HRL.3RBW
 G  HRM.3RBW
 G  NOI.3EBW
 G  NOI.3PBW
 G  NOI.3RBW
 E  WMO.3PBW
 E  WRO.3PBW
 L  XPB.3ABW
 L  XCV.3EBW
 L  XPB.3EBW
 L  XRV.3EBW
 L  X12.3EBW
 L  XCV.3PBW
 L  XRV.3PBW
 L  X12.3PBW
 L  XCV.3RBW
 L  XPB.3RBW
 L  XRV.3RBW
 L  X12.3RBW
COLUMNS
    BAL.3EBW  BAL...BW           -1.   B3E.VOBW            1.
    BAL.3EBW  XCV.3EBW            6.   XRV.3EBW            4.
    BAL.3EBW  X12.3EBW          50.3   HRL.3EBW      98.70277
    BAL.3EBW  HML.3EBW      94.63568   HRM.3EBW     102.02191
    BAL.3EBW  HMM.3EBW      98.08976   HRH.3EBW      103.0581
    BAL.3EBW  HMH.3EBW      99.18559
    BHC.3EBW  BHC...BW           -1.   B3E.VOBW            1.
    BHC.3EBW  XCV.3EBW           -2.   XRV.3EBW            .5
    BHC.3EBW  X12.3EBW         -15.6   HRL.3EBW      92.89535
    BHC.3EBW  HML.3EBW      79.40534   HRM.3EBW      94.57094
    BHC.3EBW  HMM.3EBW      81.47009   HRH.3EBW      95.02163
    BHC.3EBW  HMH.3EBW      82.04308
    BLC.3EBW  BLC...BW           -1.   B3E.VOBW            1.
    BLC.3EBW  XCV.3EBW            7.   XRV.3EBW           4.5
    BLC.3EBW  X12.3EBW          57.9   HRL.3EBW      95.38345
    BLC.3EBW  HML.3EBW      80.37873   HRM.3EBW      97.97965
    BLC.3EBW  HMM.3EBW      83.22026   HRH.3EBW      98.64634
    BLC.3EBW  HMH.3EBW       83.9937
    BLV.3EBW  BLV...BW           -1.   B3E.VOBW            1.
    BLV.3EBW  XCV.3EBW           14.   XRV.3EBW           7.2
    BLV.3EBW  X12.3EBW         102.3   HRL.3EBW       82.8797
    BLV.3EBW  HML.3EBW      80.36789   HRM.3EBW      87.33298
    BLV.3EBW  HMM.3EBW       84.5191   HRH.3EBW      88.46612
    BLV.3EBW  HMH.3EBW      85.61385
    BN4.3EBW  BN4...BW           -1.   B3E.VOBW            1.
    BN4.3EBW  XCV.3EBW           80.   XRV.3EBW           70.
    BN4.3EBW  X12.3EBW          113.   HRL.3EBW      97.32996
    BN4.3EBW  HML.3EBW      92.71594   HRM.3EBW        100.65
    BN4.3EBW  HMM.3EBW      96.86628   HRH.3EBW     101.66321
    BN4.3EBW  HMH.3EBW      98.06433
    BP8.3EBW  BP8...BW           -1.   B3E.VOBW            1.
    BP8.3EBW  XCV.3EBW            4.   XRV.3EBW           3.6
    BP8.3EBW  X12.3EBW          28.9   HRL.3EBW     101.17309
    BP8.3EBW  HML.3EBW      90.03844   HRM.3EBW     102.21363
    BP8.3EBW  HMM.3EBW      91.26611   HRH.3EBW     102.51818
    BP8.3EBW  HMH.3EBW      91.62642
    BTO.3EBW  BTO...BW           -1.   B3E.VOBW            1.
    BTO.3EBW  XCV.3EBW           -1.   XRV.3EBW           1.2
    BTO.3EBW  X12.3EBW            5.   HRL.3EBW     105.47666
    BTO.3EBW  HML.3EBW      89.10432   HRM.3EBW     106.21918
    BTO.3EBW  HMM.3EBW      90.14887   HRH.3EBW     106.46719
    BTO.3EBW  HMH.3EBW      90.49629
    BAL.3PBW  BAL...BW           -1.   B3P.VOBW            1.
    BAL.3PBW  XCV.3PBW            6.   XRV.3PBW            4.
    BAL.3PBW  X12.3PBW          50.3   WRO.3PBW      96.13556
    BAL.3PBW  WMO.3PBW      91.96313
    BHC.3PBW  BHC...BW           -1.   B3P.VOBW            1.
    BHC.3PBW  XCV.3PBW           -2.   XRV.3PBW            .5
    BHC.3PBW  X12.3PBW         -15.6   WRO.3PBW      90.99637
    BHC.3PBW  WMO.3PBW      78.09095
    BLC.3PBW  BLC...BW           -1.   B3P.VOBW            1.
    BLC.3PBW  XCV.3PBW            7.   XRV.3PBW           4.5
    BLC.3PBW  X12.3PBW          57.9   WRO.3PBW      93.95665
    BLC.3PBW  WMO.3PBW      80.74635
    BLV.3PBW  BLV...BW           -1.   B3P.VOBW            1.
    BLV.3PBW  XCV.3PBW           14.   XRV.3PBW           7.2
    BLV.3PBW  X12.3PBW         102.3   WRO.3PBW      79.78002
    BLV.3PBW  WMO.3PBW      77.37441
    BN4.3PBW  BN4...BW           -1.   B3P.VOBW            1.
    BN4.3PBW  XCV.3PBW           80.   XRV.3PBW           70.
    BN4.3PBW  X12.3PBW          113.   WRO.3PBW      94.11062
    BN4.3PBW  WMO.3PBW      88.35436
    BP8.3PBW  BP8...BW           -1.   B3P.VOBW            1.
    BP8.3PBW  XCV.3PBW            4.   XRV.3PBW           3.6
    BP8.3PBW  X12.3PBW          28.9   WRO.3PBW      99.83178
    BP8.3PBW  WMO.3PBW      88.58029
    BTO.3PBW  BTO...BW           -1.   B3P.VOBW            1.
    BTO.3PBW  XCV.3PBW           -1.   XRV.3PBW           1.2
    BTO.3PBW  X12.3PBW            5.   WRO.3PBW     105.07558
    BTO.3PBW  WMO.3PBW      88.18188
    BAL.3RBW  BAL...BW           -1.   B3R.VOBW            1.
    BAL.3RBW  XCV.3RBW            6.   XRV.3RBW            4.
    BAL.3RBW  X12.3RBW          50.3   HRL.3RBW      99.19039
    BAL.3RBW  HML.3RBW      95.17073   HRM.3RBW      101.0885
    BAL.3RBW  HMM.3RBW      97.11016   HRH.3RBW      103.0581
    BAL.3RBW  HMH.3RBW      99.18559
    BHC.3RBW  BHC...BW           -1.   B3R.VOBW            1.
    BHC.3RBW  XCV.3RBW           -2.   XRV.3RBW            .5
    BHC.3RBW  X12.3RBW         -15.6   HRL.3RBW      93.16124
    BHC.3RBW  HML.3RBW      79.72867   HRM.3RBW      94.14769
    BHC.3RBW  HMM.3RBW      80.94047   HRH.3RBW      95.02163
    BHC.3RBW  HMH.3RBW      82.04308
    BLC.3RBW  BLC...BW           -1.   B3R.VOBW            1.
    BLC.3RBW  XCV.3RBW            7.   XRV.3RBW           4.5
    BLC.3RBW  X12.3RBW          57.9   HRL.3RBW      95.80861
    BLC.3RBW  HML.3RBW      80.82888   HRM.3RBW      97.34183
    BLC.3RBW  HMM.3RBW      82.49926   HRH.3RBW      98.64634
    BLC.3RBW  HMH.3RBW       83.9937
    BLV.3RBW  BLV...BW           -1.   B3R.VOBW            1.
    BLV.3RBW  XCV.3RBW           14.   XRV.3RBW           7.2
    BLV.3RBW  X12.3RBW         102.3   HRL.3RBW      83.61375
    BLV.3RBW  HML.3RBW      81.03825   HRM.3RBW      86.24515
    BLV.3RBW  HMM.3RBW      83.48458   HRH.3RBW      88.46612
    BLV.3RBW  HMH.3RBW      85.61385
    BN4.3RBW  BN4...BW           -1.   B3R.VOBW            1.
    BN4.3RBW  XCV.3RBW           80.   XRV.3RBW           70.
    BN4.3RBW  X12.3RBW          113.   HRL.3RBW      97.86876
    BN4.3RBW  HML.3RBW      93.41749   HRM.3RBW      99.77765
    BN4.3RBW  HMM.3RBW      95.86635   HRH.3RBW     101.66321
    BN4.3RBW  HMH.3RBW      98.06433
    BP8.3RBW  BP8...BW           -1.   B3R.VOBW            1.
    BP8.3RBW  XCV.3RBW            4.   XRV.3RBW           3.6
    BP8.3RBW  X12.3RBW          28.9   HRL.3RBW     101.32905
    BP8.3RBW  HML.3RBW      90.22411   HRM.3RBW     101.93754
    BP8.3RBW  HMM.3RBW      90.94112   HRH.3RBW     102.51818
    BP8.3RBW  HMH.3RBW      91.62642
    BTO.3RBW  BTO...BW           -1.   B3R.VOBW            1.
    BTO.3RBW  XCV.3RBW           -1.   XRV.3RBW           1.2
    BTO.3RBW  X12.3RBW            5.   HRL.3RBW     105.58392
    BTO.3RBW  HML.3RBW      89.25587   HRM.3RBW      106.0019
    BTO.3RBW  HMM.3RBW      89.84584   HRH.3RBW     106.46719
    BTO.3RBW  HMH.3RBW      90.49629
    D3T...BW  B3T...BW           -1.   FAT7..J.         -16.5
    EAL...BW  BAL...BW            1.
    EHC...BW  BHC...BW            1.
    ELC...BW  BLC...BW            1.
    ELV...BW  BLV...BW            1.
    EN4...BW  BN4...BW            1.   FAT7..J.           12.
    EP8...BW  BP8...BW            1.
    ETO...BW  BTO...BW            1.   FAT7..J.           16.
    M3..3TBW  B3T...BW            1.   B3E...BW          -.29
    M3..3TBW  B3P...BW          -.17   B3R...BW          -.54
    QPB73EBW  FAT7..J.        .08757   XPB.3EBW            1.
    QPB73EBW  HRL.3EBW       2.52143   HML.3EBW       3.42918
    QPB73EBW  HRM.3EBW       1.54954   HMM.3EBW       1.55751
    QPB73EBW  HRH.3EBW       1.27141   HMH.3EBW       1.23842
    QPB73EBW  XPB.3ABW            1.
    QVO73EBW  B3E...BW            1.   B3E.VOBW           -1.
    QVO73EBW  XCV.3EBW          -16.   XRV.3EBW          -12.
    QVO73EBW  X12.3EBW          -61.   NOI.3EBW       -107.52
    QVO73EBW  XPB.3EBW          -1.7   HRL.3EBW      -1.00857
    QVO73EBW  HML.3EBW      -1.37167   HRM.3EBW       -2.0144
    QVO73EBW  HMM.3EBW      -2.02477   HRH.3EBW      -2.16139
    QVO73EBW  HMH.3EBW      -2.10531   XPB.3ABW          -1.5
    QVO73PBW  B3P...BW            1.   B3P.VOBW           -1.
    QVO73PBW  XCV.3PBW          -16.   XRV.3PBW          -12.
    QVO73PBW  X12.3PBW          -61.   NOI.3PBW        -97.41
    QVO73PBW  XPB.3ABW          -1.5
    QPB73RBW  FAT7..J.        .08757   XPB.3RBW            1.
    QPB73RBW  HRL.3RBW       4.31949   HML.3RBW       4.41873
    QPB73RBW  HRM.3RBW       2.62696   HMM.3RBW       2.74531
    QPB73RBW  HRH.3RBW       1.64391   HMH.3RBW       1.75028
    QPB73RBW  XPB.3ABW            1.
    QVO73RBW  B3R...BW            1.   B3R.VOBW           -1.
    QVO73RBW  XCV.3RBW          -16.   XRV.3RBW          -12.
    QVO73RBW  X12.3RBW          -61.   NOI.3RBW         -98.5
    QVO73RBW  XPB.3RBW          -1.7   HRL.3RBW      -2.15975
    QVO73RBW  HML.3RBW      -2.20937   HRM.3RBW      -2.62696
    QVO73RBW  HMM.3RBW      -2.74531   HRH.3RBW      -2.79464
    QVO73RBW  HMH.3RBW      -2.97548   XPB.3ABW          -1.5
    WMO73EBW  NOI.3EBW           .73   HML.3EBW           -1.
    WMO73EBW  HMM.3EBW           -1.   HMH.3EBW           -1.
    WRO73EBW  NOI.3EBW           .41   HRL.3EBW           -1.
    WRO73EBW  HRM.3EBW           -1.   HRH.3EBW           -1.
    WMO73PBW  WMO.3PBW           -1.   NOI.3PBW           .84
    WRO73PBW  WRO.3PBW           -1.   NOI.3PBW           .27
    WMO73RBW  NOI.3RBW           .81   HML.3RBW           -1.
    WMO73RBW  HMM.3RBW           -1.   HMH.3RBW           -1.
    WRO73RBW  NOI.3RBW           .31   HRL.3RBW           -1.
    WRO73RBW  HRM.3RBW           -1.   HRH.3RBW           -1.
RHS
BOUNDS
 UP 77BOUND   BHC.3EBW           10.
 UP 77BOUND   D3T...BW          200.
 UP 77BOUND   EAL...BW           10.
 UP 77BOUND   EHC...BW           20.
 UP 77BOUND   ELC...BW           25.
 UP 77BOUND   ELV...BW           12.
 UP 77BOUND   EN4...BW          100.
 UP 77BOUND   EP8...BW           35.
 UP 77BOUND   ETO...BW            5.
ENDATA
