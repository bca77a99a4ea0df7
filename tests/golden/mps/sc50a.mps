NAME          SC50A
ROWS
 N  MAXIM
 L  ROW00001
 L  ROW00002
 L  ROW00003
 E  ROW00004
 E  ROW00005
 E  ROW00006
 E  ROW00007
 L  ROW00008
 L  ROW00009
 L  ROW00010
 L  ROW00011
 L  ROW00012
 L  ROW00013
 E  ROW00014
 E  ROW00015
 E  ROW00016
 E  ROW00017
 E  ROW00018
 L  ROW00019
 L  ROW00020
 L  ROW00021
 L  ROW00022
 L  ROW00023
 L  ROW00024
 E  ROW00025
 E  ROW00026
 E  ROW00027
 E  ROW00028
 E  ROW00029
 L  ROW00030
 L  ROW00031
 L  ROW00032
 L  ROW00033
 L  ROW00034
 L  ROW00035
 E  ROW00036
 E  ROW00037
 E  ROW00038
 E  ROW00039
 E  ROW00040
 L  ROW00041
 L  ROW00042
 L  ROW00043
 L  ROW00044
 L  ROW00045
 L  ROW00046
 E  ROW00047
 L  ROW00048
 L  ROW00049
 L  ROW00050
COLUMNS
    COL00001  ROW00001            2.   ROW00002            1.
    COL00001  ROW00005           -1.
    COL00002  ROW00001            1.   ROW00002            2.
    COL00002  ROW00006           -1.
    COL00003  ROW00001           1.5   ROW00002           1.5
    COL00003  ROW00007           -1.
    COL00004  MAXIM              -1.   ROW00004            1.
    COL00004  ROW00014           1.1
    COL00005  ROW00004           -1.   ROW00013            1.
    COL00005  ROW00015            1.
    COL00006  ROW00005            1.   ROW00008           -1.
    COL00006  ROW00016           -1.
    COL00007  ROW00006            1.   ROW00009           -1.
    COL00007  ROW00017           -1.
    COL00008  ROW00007            1.   ROW00010           -1.
    COL00008  ROW00018           -1.
    COL00009  ROW00008            1.   ROW00011           -1.
    COL00010  ROW00009            1.   ROW00012           -1.
    COL00011  ROW00010            1.   ROW00013           -1.
    COL00012  ROW00011            2.   ROW00012            1.
    COL00012  ROW00016           -1.
    COL00013  ROW00011            1.   ROW00012            2.
    COL00013  ROW00017           -1.
    COL00014  ROW00011           1.5   ROW00012           1.5
    COL00014  ROW00018           -1.
    COL00015  ROW00014           -1.   ROW00015            1.
    COL00015  ROW00025           1.1
    COL00016  ROW00015           -1.   ROW00024            1.
    COL00016  ROW00026            1.
    COL00017  ROW00016            1.   ROW00019           -1.
    COL00017  ROW00027           -1.
    COL00018  ROW00017            1.   ROW00020           -1.
    COL00018  ROW00028           -1.
    COL00019  ROW00018            1.   ROW00021           -1.
    COL00019  ROW00029           -1.
    COL00020  ROW00019            1.   ROW00022           -1.
    COL00021  ROW00020            1.   ROW00023           -1.
    COL00022  ROW00021            1.   ROW00024           -1.
    COL00023  ROW00022            2.   ROW00023            1.
    COL00023  ROW00027           -1.
    COL00024  ROW00022            1.   ROW00023            2.
    COL00024  ROW00028           -1.
    COL00025  ROW00022           1.5   ROW00023           1.5
    COL00025  ROW00029           -1.
    COL00026  ROW00025           -1.   ROW00026            1.
    COL00026  ROW00036           1.1
    COL00027  ROW00026           -1.   ROW00035            1.
    COL00027  ROW00037            1.
    COL00028  ROW00027            1.   ROW00030           -1.
    COL00028  ROW00038           -1.
    COL00029  ROW00028            1.   ROW00031           -1.
    COL00029  ROW00039           -1.
    COL00030  ROW00029            1.   ROW00032           -1.
    COL00030  ROW00040           -1.
    COL00031  ROW00030            1.   ROW00033           -1.
    COL00032  ROW00031            1.   ROW00034           -1.
    COL00033  ROW00032            1.   ROW00035           -1.
    COL00034  ROW00033            2.   ROW00034            1.
    COL00034  ROW00038           -1.
    COL00035  ROW00033            1.   ROW00034            2.
    COL00035  ROW00039           -1.
    COL00036  ROW00033           1.5   ROW00034           1.5
    COL00036  ROW00040           -1.
    COL00037  ROW00036           -1.   ROW00037            1.
    COL00037  ROW00047           1.1
    COL00038  ROW00037           -1.   ROW00046            1.
    COL00039  ROW00038            1.   ROW00041           -1.
    COL00040  ROW00039            1.   ROW00042           -1.
    COL00041  ROW00040            1.   ROW00043           -1.
    COL00042  ROW00041            1.   ROW00044           -1.
    COL00043  ROW00042            1.   ROW00045           -1.
    COL00044  ROW00043            1.   ROW00046           -1.
    COL00045  ROW00044            2.   ROW00045            1.
    COL00045  ROW00048           -.8   ROW00049            .1
    COL00046  ROW00044            1.   ROW00045            2.
    COL00046  ROW00048            .1   ROW00049           -.8
    COL00047  ROW00044           1.5   ROW00045           1.5
    COL00047  ROW00048           .15   ROW00049           .15
    COL00047  ROW00050           -1.
    COL00048  ROW00047           -1.   ROW00050            1.
RHS
    CONST     ROW00001          170.   ROW00002          130.
    CONST     ROW00011          170.   ROW00012          130.
    CONST     ROW00022          170.   ROW00023          130.
    CONST     ROW00033          170.   ROW00034          130.
    CONST     ROW00044          170.   ROW00045          130.
ENDATA
